"""Scope row f3: the binary R1CS / witness interchange files (zukelang_amd/r1cs_file.py) -- the CSR form of the
reference's `Gate {lhs; l; r}` affine maps (src/lib/zk/circuit.ml:73-75).  CPU only."""
import os

import numpy as np
import pytest

from zukelang_amd import r1cs as RC
from zukelang_amd import r1cs_file as RF

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
README_VARS = [("ONE", 1), ("c", 4), ("c", 5), ("input", 3), ("v", 6)]


def _golden(name):
    return bytes.fromhex(open(os.path.join(GOLDEN, name)).read().strip())


def _same(a, b):
    return (a.n, a.m) == (b.n, b.m) and bytes(a.mid) == bytes(b.mid) and all(
        np.array_equal(x.ptr, y.ptr) and np.array_equal(x.col, y.col) and bytes(x.val) == bytes(y.val)
        for x, y in ((a.L, b.L), (a.R, b.R), (a.O, b.O)))


def test_writer_reproduces_the_golden_readme_files(tmp_path):
    cs, w = RC.readme_circuit(3)
    p, q = str(tmp_path / "c.r1cs"), str(tmp_path / "c.wit")
    RF.write_r1cs(p, cs, README_VARS)
    RF.write_witness(q, w)
    assert open(p, "rb").read() == _golden("readme_circuit.r1cs.hex")
    assert open(q, "rb").read() == _golden("readme_circuit_x3.wit.hex")


def test_reader_recovers_the_readme_circuit_from_the_golden_files():
    cs, names = RF.read_r1cs(_golden("readme_circuit.r1cs.hex"))
    ref, w = RC.readme_circuit(3)
    assert _same(cs, ref) and names == README_VARS
    sol = RF.read_witness(_golden("readme_circuit_x3.wit.hex"))
    assert RC.fr_ints(sol) == w and cs.check(RC.fr_ints(sol))


@pytest.mark.parametrize("n", [2, 64, 5000])
def test_round_trip_of_the_benchmark_family(tmp_path, n):
    cs, w = RC.iterated_cubic(n, 99)
    p, q = str(tmp_path / "c.r1cs"), str(tmp_path / "c.wit")
    RF.write_r1cs(p, cs)
    RF.write_witness(q, w)
    back, names = RF.read_r1cs(p)
    assert _same(back, cs) and len(names) == cs.m
    assert RC.fr_ints(RF.read_witness(q)) == w
    # sections are 8-byte aligned so a zk_csr can point straight into an mmap of the file
    assert os.path.getsize(p) % 8 == 0


def test_malformed_files_are_rejected(tmp_path):
    good = bytearray(_golden("readme_circuit.r1cs.hex"))
    with pytest.raises(ValueError):
        RF.read_r1cs(bytes(good[:-8]))                       # truncated
    with pytest.raises(ValueError):
        RF.read_r1cs(bytes(good) + b"\x00" * 8)              # trailing bytes
    bad = bytearray(good); bad[0] ^= 1
    with pytest.raises(ValueError):
        RF.read_r1cs(bytes(bad))                             # magic
    bad = bytearray(good); bad[-1] = 0xFF
    with pytest.raises(ValueError):
        RF.read_r1cs(bytes(bad))                             # coefficient >= r
    # a column index >= m
    cs, _ = RC.readme_circuit(3)
    cs.L.col[0] = 7
    p = str(tmp_path / "bad.r1cs")
    RF.write_r1cs(p, cs, README_VARS)
    with pytest.raises(ValueError):
        RF.read_r1cs(p)
    # variables out of Var.compare order
    cs, _ = RC.readme_circuit(3)
    RF.write_r1cs(p, cs, list(reversed(README_VARS)))
    with pytest.raises(ValueError):
        RF.read_r1cs(p)
    # a DUPLICATE (name, id): sorted() would accept it, Var.compare order is strict
    cs, _ = RC.readme_circuit(3)
    dup = list(README_VARS)
    dup[1] = dup[0]
    RF.write_r1cs(p, cs, dup)
    with pytest.raises(ValueError):
        RF.read_r1cs(p)
    wit = bytearray(_golden("readme_circuit_x3.wit.hex"))
    with pytest.raises(ValueError):
        RF.read_witness(bytes(wit[:-1]))
    wit[-1] = 0xFF
    with pytest.raises(ValueError):
        RF.read_witness(bytes(wit))
