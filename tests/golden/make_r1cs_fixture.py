"""Generates tests/golden/readme_circuit.r1cs.hex / readme_circuit_x3.wit.hex: the README circuit `x*x*x + x + 3`
(README.md:49; hand-derived compilation) in the interchange format of zukelang_amd/r1cs_file.py,
written out BYTE BY BYTE from the format description (struct.pack here, independent of the module's writer), so the
test pins the writer and the reader against the documented layout.  Run: python tests/golden/make_r1cs_fixture.py"""
import os
import struct

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
HERE = os.path.dirname(os.path.abspath(__file__))


def fr(x):
    return (x % R).to_bytes(32, "little")


def pad8(b):
    return b + b"\x00" * (-len(b) % 8)


def main():
    # variables in Var.compare order: ("ONE",1) ("c",4) ("c",5) ("input",3) ("v",6); mids = c4, c5, input
    names = [("ONE", 1), ("c", 4), ("c", 5), ("input", 3), ("v", 6)]
    mid = bytes([0, 1, 1, 1, 0])
    # gates: c4 = input*input ; c5 = c4*input ; v6 = (c5 + input + 3 ONE) * (1 ONE)
    L = [[(3, 1)], [(1, 1)], [(0, 3), (2, 1), (3, 1)]]
    Rm = [[(3, 1)], [(3, 1)], [(0, 1)]]
    O = [[(1, 1)], [(2, 1)], [(4, 1)]]
    out = b"ZKR1CS\x00\x01" + struct.pack("<II", 1, 32) + struct.pack("<QQQQQ", 3, 5, 5, 3, 3)
    for name, vid in names:
        out += struct.pack("<II", vid, len(name)) + name.encode()
    out = pad8(out) + mid
    out = pad8(out)
    for M in (L, Rm, O):
        ptr, col, val = [0], [], b""
        for row in M:
            for k, c in row:
                col.append(k)
                val += fr(c)
            ptr.append(len(col))
        out += pad8(struct.pack("<%dI" % len(ptr), *ptr))
        out += pad8(struct.pack("<%dI" % len(col), *col))
        out += val
    open(os.path.join(HERE, "readme_circuit.r1cs.hex"), "w").write(out.hex() + "\n")
    w = [1, 9, 27, 3, 33]
    wit = b"ZKWIT\x00\x00\x01" + struct.pack("<IIQ", 1, 32, 5) + b"".join(fr(x) for x in w)
    open(os.path.join(HERE, "readme_circuit_x3.wit.hex"), "w").write(wit.hex() + "\n")


if __name__ == "__main__":
    main()
