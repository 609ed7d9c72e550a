"""Binary R1CS / witness interchange files (scope row f3).

The reference has no circuit file format: a circuit is an OCaml value `Circuit.t` whose gates are
`Gate.t = { lhs : Affine.t; l : Affine.t; r : Affine.t }` with `lhs = l * r`
(src/lib/zk/circuit.ml:73-75), `Affine.t = F.t Var.Map.t`, `Var.t = string * int` (src/lib/zk/var.ml:4).
These two files are what an OCaml host writes so that this library (or any other prover) can take the
circuit and a solution WITHOUT linking OCaml: the same three CSR matrices `zk_groth16_pk_upload` takes
(include/zkmi355x.h `zk_csr`), laid out so a reader can mmap the file and point a `zk_csr` into it.

All integers little-endian; every section starts on an 8-byte boundary (zero padding).

  .r1cs   magic  "ZKR1CS\\0\\1"                            8 B
          u32 version = 1 | u32 field_bytes = 32          (Fr: 32 B little-endian canonical integer < r)
          u64 n   gates, ids = position in Gate.Set.elements order (QAP.ml:22)
          u64 m   variables, index = position in Var.compare order (var.ml:8,42), ONE included
          u64 nnz_L | u64 nnz_R | u64 nnz_O
          vars    m x { u32 id ; u32 name_len ; name bytes }   then padding       -- Var.t = (name, id)
          mid     m x u8, 1 <=> the variable is in circuit.mids (circuit.ml:108-113)   then padding
          3 x matrix, in the order L (gate.l, the reference's `v`), R (gate.r, `w`), O (gate.lhs, `y`):
              row_ptr (n+1) x u32 | padding | col nnz x u32 | padding | val nnz x 32 B
              row g lists the bindings of the gate's affine form in Var.Map order (ascending variable index)
  .wit    magic  "ZKWIT\\0\\0\\1"                           8 B
          u32 version = 1 | u32 field_bytes = 32 | u64 m
          m x 32 B   the solution `Fr.t Var.Map.t` in key order (the `sol` argument of Protocol.S.prove,
                     src/lib/zk/protocol.mli:19-23)

`write_r1cs` / `read_r1cs` and `write_witness` / `read_witness` are the whole API; `read_r1cs` validates
everything `zk_groth16_pk_upload` would otherwise have to reject (monotone row_ptr, column range, values < r).
INTEGRATION.md section 7 shows the OCaml writer.
"""
import struct

import numpy as np

from .r1cs import FR_MODULUS, Matrix, R1CS

R1CS_MAGIC = b"ZKR1CS\x00\x01"
WIT_MAGIC = b"ZKWIT\x00\x00\x01"
VERSION = 1


def _pad8(buf):
    buf += b"\x00" * (-len(buf) % 8)


def _check_fr(vals, what):
    a = np.frombuffer(bytes(vals), dtype="<u8").reshape(-1, 4)
    if a.size == 0:
        return
    mod = [(FR_MODULUS >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
    # lexicographic compare from the top word down
    lt = np.zeros(len(a), dtype=bool)
    eq = np.ones(len(a), dtype=bool)
    for i in (3, 2, 1, 0):
        lt |= eq & (a[:, i] < np.uint64(mod[i]))
        eq &= a[:, i] == np.uint64(mod[i])
    if not bool(lt.all()):
        raise ValueError("%s: a field element is >= r" % what)


def write_r1cs(path, circuit: R1CS, var_names=None):
    """var_names: list of m (name, id) pairs in Var.compare order; default ("v", k+1)."""
    n, m = circuit.n, circuit.m
    if var_names is None:
        var_names = [("v", k + 1) for k in range(m)]
    if len(var_names) != m:
        raise ValueError("var_names must list all m variables")
    out = bytearray(R1CS_MAGIC)
    out += struct.pack("<II", VERSION, 32)
    mats = (circuit.L, circuit.R, circuit.O)
    out += struct.pack("<QQQQQ", n, m, *[int(M.ptr[n]) for M in mats])
    for name, vid in var_names:
        nb = name.encode("latin-1") if isinstance(name, str) else bytes(name)
        out += struct.pack("<II", vid, len(nb)) + nb
    _pad8(out)
    out += bytes(np.ascontiguousarray(circuit.mid, dtype=np.uint8))
    _pad8(out)
    for M in mats:
        nnz = int(M.ptr[n])
        out += np.ascontiguousarray(M.ptr, dtype="<u4").tobytes()
        _pad8(out)
        out += np.ascontiguousarray(M.col[:nnz], dtype="<u4").tobytes()
        _pad8(out)
        out += bytes(M.val[:32 * nnz])
    with open(path, "wb") as f:
        f.write(bytes(out))


def read_r1cs(path):
    """-> (R1CS, var_names).  Raises ValueError on any malformed section."""
    data = open(path, "rb").read() if isinstance(path, str) else bytes(path)
    if data[:8] != R1CS_MAGIC:
        raise ValueError("not a .r1cs file (bad magic)")
    ver, fb = struct.unpack_from("<II", data, 8)
    if ver != VERSION or fb != 32:
        raise ValueError("unsupported .r1cs version / field size")
    n, m, *nnz = struct.unpack_from("<QQQQQ", data, 16)
    if n < 1 or m < 1 or n >= 1 << 31 or m >= 1 << 31:
        raise ValueError(".r1cs: bad n / m")
    pos = 56
    names = []
    for _ in range(m):
        if pos + 8 > len(data):
            raise ValueError(".r1cs: truncated variable table")
        vid, ln = struct.unpack_from("<II", data, pos)
        pos += 8
        if pos + ln > len(data):
            raise ValueError(".r1cs: truncated variable name")
        names.append((data[pos:pos + ln].decode("latin-1"), vid))      # byte order = code-point order: String.compare
        pos += ln
    if any(a >= b for a, b in zip(names, names[1:])):
        raise ValueError(".r1cs: variables are not in STRICT Var.compare order (duplicate or misordered (name, id))")       # polymorphic compare on (string * int)
    pos += -pos % 8
    if pos + m > len(data):
        raise ValueError(".r1cs: truncated mid flags")
    mid = np.frombuffer(data, dtype=np.uint8, count=m, offset=pos).copy()
    if mid.max(initial=0) > 1:
        raise ValueError(".r1cs: mid flags must be 0 / 1")
    pos += m
    pos += -pos % 8
    mats = []
    for z in nnz:
        need = 4 * (n + 1)
        if pos + need > len(data):
            raise ValueError(".r1cs: truncated row_ptr")
        ptr = np.frombuffer(data, dtype="<u4", count=n + 1, offset=pos).astype(np.uint32)
        pos += need
        pos += -pos % 8
        if int(ptr[0]) != 0 or int(ptr[n]) != z or bool((np.diff(ptr.astype(np.int64)) < 0).any()):
            raise ValueError(".r1cs: row_ptr is not a monotone prefix sum ending at nnz")
        if pos + 4 * z > len(data):
            raise ValueError(".r1cs: truncated col")
        col = np.frombuffer(data, dtype="<u4", count=z, offset=pos).astype(np.uint32)
        pos += 4 * z
        pos += -pos % 8
        if z and int(col.max()) >= m:
            raise ValueError(".r1cs: column index out of range")
        if pos + 32 * z > len(data):
            raise ValueError(".r1cs: truncated val")
        val = np.frombuffer(data, dtype=np.uint8, count=32 * z, offset=pos).copy()
        pos += 32 * z
        _check_fr(val, ".r1cs")
        for g in range(n) if n <= 4096 else ():                 # small files: Var.Map order inside a row
            row = col[ptr[g]:ptr[g + 1]]
            if len(row) > 1 and bool((np.diff(row.astype(np.int64)) <= 0).any()):
                raise ValueError(".r1cs: row %d is not in ascending variable order" % g)
        mats.append(Matrix(ptr, col, val))
    if pos != len(data):
        raise ValueError(".r1cs: trailing bytes")
    return R1CS(int(n), int(m), mats[0], mats[1], mats[2], mid), names


def write_witness(path, values):
    """values: m Python ints, or 32*m bytes already in the Fr byte format."""
    if isinstance(values, (bytes, bytearray, np.ndarray)):
        b = bytes(values)
    else:
        b = b"".join(int(v % FR_MODULUS).to_bytes(32, "little") for v in values)
    if len(b) % 32:
        raise ValueError("witness bytes must be a multiple of 32")
    with open(path, "wb") as f:
        f.write(WIT_MAGIC + struct.pack("<IIQ", VERSION, 32, len(b) // 32) + b)


def read_witness(path):
    """-> uint8 array of 32*m bytes (what Groth16.prove_rs / zk_groth16_prove take as `sol`)."""
    data = open(path, "rb").read() if isinstance(path, str) else bytes(path)
    if data[:8] != WIT_MAGIC:
        raise ValueError("not a .wit file (bad magic)")
    ver, fb, m = struct.unpack_from("<IIQ", data, 8)
    if ver != VERSION or fb != 32 or len(data) != 24 + 32 * m:
        raise ValueError(".wit: bad header or length")
    vals = np.frombuffer(data, dtype=np.uint8, count=32 * m, offset=24).copy()
    _check_fr(vals, ".wit")
    return vals
