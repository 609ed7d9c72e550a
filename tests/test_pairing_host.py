"""The product's host-side pairing (zk_pairing_product / zk_pairing_check, scope row f1) against the
oracle's Python big-int pairing: same GT element coefficient by coefficient, bilinearity, and the
rejection of points outside the curve / subgroup.  No GPU involved."""
import ctypes as C

import pytest

from oracle import pyref as P
from zukelang_amd import _lib


def gt_bytes(f):
    """oracle Fp12 -> the library's GT encoding: c0.c0.a, c0.c0.b, c0.c1.a, ... c1.c2.b, 48 B big-endian."""
    out = b""
    for six in (f.c0, f.c1):
        for c in (six.c0, six.c1, six.c2):
            out += c.a.to_bytes(48, "big") + c.b.to_bytes(48, "big")
    return out


def product(pairs):
    g1 = b"".join(P.g1_to_bytes(p) for p, _ in pairs)
    g2 = b"".join(P.g2_to_bytes(q) for _, q in pairs)
    out = C.create_string_buffer(576)
    _lib.check(_lib.lib().zk_pairing_product(g1, g2, C.c_size_t(len(pairs)), out))
    return out.raw


def check(pairs):
    g1 = b"".join(P.g1_to_bytes(p) for p, _ in pairs)
    g2 = b"".join(P.g2_to_bytes(q) for _, q in pairs)
    r = C.c_int(-1)
    _lib.check(_lib.lib().zk_pairing_check(g1, g2, C.c_size_t(len(pairs)), C.byref(r)))
    return bool(r.value)


def test_gt_element_matches_the_oracle_pairing():
    a, b = 0x1234567 * 7919, 0xfedcba987654321
    p, q = P.pt_mul(P.G1, a), P.pt_mul(P.G2, b)
    assert product([(p, q)]) == gt_bytes(P.pairing(p, q))
    # the empty product and pairings with the identity are 1
    one = gt_bytes(P.FP12_ONE)
    assert product([]) == one and product([(None, q)]) == one and product([(p, None)]) == one


def test_bilinearity_and_products():
    a, b = 5, 11
    assert check([(P.pt_mul(P.G1, a), P.pt_mul(P.G2, b)), (P.pt_neg(P.pt_mul(P.G1, a * b)), P.G2)])
    assert not check([(P.pt_mul(P.G1, a), P.pt_mul(P.G2, b)), (P.pt_neg(P.pt_mul(P.G1, a * b + 1)), P.G2)])
    s = [3, 1 << 200, P.R - 2]
    lhs = [(P.pt_mul(P.G1, x), P.pt_mul(P.G2, x + 1)) for x in s]
    total = sum(x * (x + 1) for x in s) % P.R
    assert check(lhs + [(P.pt_neg(P.pt_mul(P.G1, total)), P.G2)])


def test_pairing_order_and_the_eip2537_identity():
    """e(G1, G2) has order r (not 1), e(G1,G2) e(G1,-G2) = 1 (the EIP-2537 pairing vector), e(2 G1, 3 G2) = e(G1,G2)^6 --
    on the product's host pairing and on the oracle's big-int pairing."""
    e = P.pairing(P.G1, P.G2)
    assert e != P.FP12_ONE
    assert e.pow(P.R) == P.FP12_ONE
    assert product([(P.G1, P.G2)]) == gt_bytes(e) and product([(P.G1, P.G2)]) != gt_bytes(P.FP12_ONE)
    assert check([(P.G1, P.G2), (P.G1, P.pt_neg(P.G2))]) and check([(P.G1, P.G2), (P.pt_neg(P.G1), P.G2)])
    assert product([(P.pt_mul(P.G1, 2), P.pt_mul(P.G2, 3))]) == gt_bytes(e.pow(6)) == product([(P.pt_mul(P.G1, 6), P.G2)])
    assert product([(P.pt_mul(P.G1, P.R - 1), P.G2), (P.G1, P.G2)]) == gt_bytes(P.FP12_ONE)


def test_rejects_points_off_the_curve_or_outside_the_subgroup():
    good2 = P.g2_to_bytes(P.G2)
    bad = bytearray(P.g1_to_bytes(P.G1))
    bad[95] ^= 1                                                  # y changed: not on the curve
    out = C.create_string_buffer(576)
    assert _lib.lib().zk_pairing_product(bytes(bad), good2, C.c_size_t(1), out) != 0
    # a point of E(Fp) outside the r-torsion: x = 4 gives a valid y (cofactor part)
    x = 0
    while True:
        x += 1
        y2 = (x ** 3 + 4) % P.P
        y = pow(y2, (P.P + 1) // 4, P.P)
        if y * y % P.P == y2:
            pt = (P.Fp1(x), P.Fp1(y))
            if P.pt_mul(pt, P.R) is not None:
                break
    assert _lib.lib().zk_pairing_product(P.g1_to_bytes(pt), good2, C.c_size_t(1), out) != 0
    comp = bytearray(P.g1_to_bytes(P.G1)); comp[0] |= 0x80          # compressed flag on a 96-byte encoding
    assert _lib.lib().zk_pairing_product(bytes(comp), good2, C.c_size_t(1), out) != 0


# ---- whole verifiers (host code: no GPU): keys and proofs made by the ORACLE, checked by the product
import oracle_lib as O  # noqa: E402
from zukelang_amd import r1cs as RC  # noqa: E402


def frs(xs):
    return b"".join(P.fr_to_bytes(x) for x in xs)


def test_groth16_verify_accepts_the_oracle_proof_and_rejects_changes():
    cs, w = RC.readme_circuit(3)
    csr = [O.CSR(M.ptr, M.col, M.val) for M in (cs.L, cs.R, cs.O)]
    st = P.fr_stream(0x5EED0002)
    toxic = [next(st) for _ in range(5)]
    r, s = next(st), next(st)
    q = O.QAP(cs.n, cs.m, *csr)
    pk1, pk2, vk1, vk2 = q.groth16_setup(frs(toxic), cs.mid)
    a, b, c = O.groth16_prove_trapdoor(cs.n, cs.m, *csr, cs.mid, frs(w), frs(toxic), P.fr_to_bytes(r), P.fr_to_bytes(s))
    io = [w[k] for k in range(cs.m) if not cs.mid[k]]
    n_io = len(io)
    # oracle vkey layout (orc_groth16_setup): vk1 = one1 | ltgm_io[n_io]; vk2 = one2 | gm | d
    assert len(vk1) == 96 * (1 + n_io) and len(vk2) == 3 * 192
    vk1, vk2 = vk1[96:], vk2[192:]
    out = C.create_string_buffer(576)
    _lib.check(_lib.lib().zk_pairing_product(pk1[:96], pk2[:192], C.c_size_t(1), out))     # ab = e(alpha, beta)
    ab = out.raw

    def verify(proof, io_vals):
        ok = C.c_int(-1)
        _lib.check(_lib.lib().zk_groth16_verify(ab, vk1, frs(io_vals), C.c_size_t(n_io), vk2[:192], vk2[192:], proof, C.byref(ok)))
        return bool(ok.value)

    assert verify(a + b + c, io)
    assert not verify(a + b + P.g1_to_bytes(P.pt_mul(P.g1_from_bytes(c), 2)), io)            # another C
    bad_io = list(io); bad_io[-1] = (bad_io[-1] + 1) % P.R
    assert not verify(a + b + c, bad_io)


def test_pinocchio_verify_accepts_the_oracle_proof_and_rejects_changes():
    cs, w = RC.iterated_cubic(6, 9)
    csr = [O.CSR(M.ptr, M.col, M.val) for M in (cs.L, cs.R, cs.O)]
    st = P.fr_stream(0x5EED0003)
    tox = [next(st) for _ in range(11)]
    toxic = frs(tox[:8])
    ex = O.pinocchio_keygen_exponents(None, cs.n, cs.m, *csr, cs.mid, toxic, False)
    vk1, vk2 = O.points_of_exponents_g1(ex[2]), O.points_of_exponents_g2(ex[3])
    dv, dw, dy = (P.fr_to_bytes(x) for x in tox[8:])
    proof = O.pinocchio_prove_trapdoor(cs.n, cs.m, *csr, cs.mid, frs(w), toxic, dv, dw, dy)
    io = [w[k] for k in range(cs.m) if not cs.mid[k]]

    def verify(pr, io_vals):
        ok = C.c_int(-1)
        _lib.check(_lib.lib().zk_pinocchio_verify(vk1, vk2, frs(io_vals), C.c_size_t(len(io_vals)), pr, C.byref(ok)))
        return bool(ok.value)

    assert verify(proof, io)
    assert O.pinocchio_verify(vk1, vk2, io, proof)                    # the oracle agrees
    bad = bytearray(proof)
    bad[384:480] = P.g1_to_bytes(P.pt_mul(P.g1_from_bytes(bytes(proof[384:480])), 3))      # another h
    assert not verify(bytes(bad), io)
    bad_io = list(io); bad_io[0] = (bad_io[0] + 1) % P.R
    assert not verify(proof, bad_io)
