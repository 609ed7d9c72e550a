"""R1CS interchange format for the prove path, plus the benchmark circuit families.

The reference has no R1CS file format: its circuits are OCaml values
`Circuit.Gate.Set.t` of gates `{lhs; l; r}` meaning `lhs = l * r`
(src/lib/zk/circuit.ml:73-75), with `Affine.t = F.t Var.Map.t` sparse linear forms.
QAP.build numbers the gates 0..n-1 in `Gate.Set.elements` order (src/lib/zk/QAP.ml:22) and
reads, for every variable k, the coefficient of k in l / r / lhs of each gate (QAP.ml:30-52).
That is exactly three sparse n x m matrices; we carry them as CSR:

  L (the reference's `v`, left operand), R (`w`, right operand), O (`y`, the lhs),
  rows = gate ids in Gate.Set order, columns = dense variable indices in Var.compare order
  (src/lib/zk/var.ml:8,42: polymorphic compare on (string * int)), values = Fr.

`mid[k] = 1` marks `circuit.mids` (src/lib/zk/circuit.ml:108-113); the rest is
inputs_public + outputs (groth16.ml:231).
"""
from dataclasses import dataclass

import numpy as np

FR_MODULUS = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def fr_bytes(values):
    """Iterable of Python ints -> contiguous uint8 array of 32 B little-endian canonical Fr."""
    out = bytearray()
    for v in values:
        out += int(v % FR_MODULUS).to_bytes(32, "little")
    return np.frombuffer(bytes(out), dtype=np.uint8).copy()


def fr_ints(buf):
    b = bytes(buf)
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


@dataclass
class Matrix:
    ptr: np.ndarray   # uint32 [n+1]
    col: np.ndarray   # uint32 [nnz]
    val: np.ndarray   # uint8  [nnz*32]

    @staticmethod
    def from_rows(rows):
        """rows: list over gates of {var_index: int coefficient}."""
        ptr = [0]
        col = []
        vals = []
        for row in rows:
            for k in sorted(row):
                col.append(k)
                vals.append(row[k])
            ptr.append(len(col))
        return Matrix(np.array(ptr, dtype=np.uint32), np.array(col, dtype=np.uint32), fr_bytes(vals))


    def transposed(self, cols):
        """The transpose as another CSR matrix (rows = this matrix's columns): what keygen multiplies by the Lagrange basis at tau
        (u_k(tau) = sum_g M[g][k] l_g(tau)).  Stable in the row order, so every column's entries stay in gate order."""
        n = len(self.ptr) - 1
        rows = np.repeat(np.arange(n, dtype=np.uint32), np.diff(self.ptr.astype(np.int64)))
        order = np.argsort(self.col, kind="stable")
        tptr = np.zeros(cols + 1, dtype=np.uint32)
        tptr[1:] = np.cumsum(np.bincount(self.col, minlength=cols))
        tval = np.ascontiguousarray(self.val.reshape(-1, 32)[order]).reshape(-1)
        return Matrix(tptr, np.ascontiguousarray(rows[order]), tval)


@dataclass
class R1CS:
    n: int            # gates / constraints
    m: int            # variables incl. ONE
    L: Matrix
    R: Matrix
    O: Matrix
    mid: np.ndarray   # uint8 [m]

    @property
    def n_mid(self):
        return int(self.mid.sum())

    def check(self, witness_ints):
        """lhs = l * r on every gate (host-side, Python ints; small circuits only)."""
        w = witness_ints

        def row(M, g):
            acc = 0
            for e in range(M.ptr[g], M.ptr[g + 1]):
                c = int.from_bytes(bytes(M.val[32 * e:32 * e + 32]), "little")
                acc += c * w[M.col[e]]
            return acc % FR_MODULUS
        return all(row(self.L, g) * row(self.R, g) % FR_MODULUS == row(self.O, g) for g in range(self.n))


def readme_circuit(x=3):
    """`x*x*x + x + 3` (README.md:49, src/lib/test/test.ml:195-197) as compiled by Comp
    (hand-derived in SURVEY.md 8c): variables in Var.compare order
      0 ("ONE",1)  1 ("c",4)  2 ("c",5)  3 ("input",3)  4 ("v",6)
    gates (Gate.compare on lhs): c4 = input*input ; c5 = c4*input ; v6 = (c5+input+3*ONE)*(1*ONE).
    """
    ONE, C4, C5, IN, V6 = range(5)
    L = Matrix.from_rows([{IN: 1}, {C4: 1}, {C5: 1, IN: 1, ONE: 3}])
    R = Matrix.from_rows([{IN: 1}, {IN: 1}, {ONE: 1}])
    O = Matrix.from_rows([{C4: 1}, {C5: 1}, {V6: 1}])
    mid = np.array([0, 1, 1, 1, 0], dtype=np.uint8)
    x %= FR_MODULUS
    w = [1, x * x % FR_MODULUS, x * x * x % FR_MODULUS, x, (x * x * x + x + 3) % FR_MODULUS]
    return R1CS(3, 5, L, R, O, mid), w


def iterated_cubic(n, x):
    """The synthetic benchmark family of SURVEY.md 8d: u -> u^3 + u + 3 applied n/2 times.
    Gates 2j: t_j = u_j * u_j ;  2j+1: u_{j+1} - u_j - 3*ONE = t_j * u_j.
    Variables: 0 = ONE, 1 = u_0 = x, 2+2j = t_j, 3+2j = u_{j+1}; m = n + 2.
    Public = {ONE, u_{n/2}}, mids = the rest (n of them).  Returns (R1CS, witness ints)."""
    assert n >= 2 and n % 2 == 0
    half = n // 2
    m = n + 2
    j = np.arange(half, dtype=np.uint32)
    uj = 1 + 2 * j
    tj = 2 + 2 * j
    uj1 = 3 + 2 * j
    one = int(1).to_bytes(32, "little")
    minus1 = int(FR_MODULUS - 1).to_bytes(32, "little")
    minus3 = int(FR_MODULUS - 3).to_bytes(32, "little")

    def const_vals(count, b):
        return np.tile(np.frombuffer(b, dtype=np.uint8), count)

    # L: one entry per gate: even -> u_j, odd -> t_j
    lcol = np.empty(n, dtype=np.uint32); lcol[0::2] = uj; lcol[1::2] = tj
    rcol = np.empty(n, dtype=np.uint32); rcol[0::2] = uj; rcol[1::2] = uj
    ptr1 = np.arange(n + 1, dtype=np.uint32)
    L = Matrix(ptr1.copy(), lcol, const_vals(n, one))
    R = Matrix(ptr1.copy(), rcol, const_vals(n, one))
    # O: even rows {t_j:1}; odd rows {ONE:-3, u_j:-1, u_{j+1}:1} (sorted by column)
    optr = np.zeros(n + 1, dtype=np.uint32)
    cnt = np.empty(n, dtype=np.uint32); cnt[0::2] = 1; cnt[1::2] = 3
    optr[1:] = np.cumsum(cnt)
    ocol = np.empty(int(optr[-1]), dtype=np.uint32)
    oval = np.empty((int(optr[-1]), 32), dtype=np.uint8)
    ev = optr[0:n:2]
    od = optr[1:n:2]
    ocol[ev] = tj; oval[ev] = np.frombuffer(one, dtype=np.uint8)
    ocol[od] = 0; oval[od] = np.frombuffer(minus3, dtype=np.uint8)
    ocol[od + 1] = uj; oval[od + 1] = np.frombuffer(minus1, dtype=np.uint8)
    ocol[od + 2] = uj1; oval[od + 2] = np.frombuffer(one, dtype=np.uint8)
    O = Matrix(optr, ocol, oval.reshape(-1))
    mid = np.ones(m, dtype=np.uint8); mid[0] = 0; mid[m - 1] = 0
    # witness by forward evaluation
    w = [0] * m
    w[0] = 1
    u = x % FR_MODULUS
    w[1] = u
    for k in range(half):
        t = u * u % FR_MODULUS
        u1 = (t * u + u + 3) % FR_MODULUS
        w[2 + 2 * k] = t
        w[3 + 2 * k] = u1
        u = u1
    return R1CS(n, m, L, R, O, mid), w


def splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return state, z ^ (z >> 31)


def fr_stream(seed):
    """8 successive splitmix64 outputs -> 512 bits -> reduced mod r."""
    st = seed
    while True:
        v = 0
        for _ in range(8):
            st, o = splitmix64(st)
            v = (v << 64) | o
        yield v % FR_MODULUS


def random_fr_bytes(count, seed):
    """count uniformly distributed Fr elements as a uint8 array (numpy PCG, rejection-free:
    256 random bits masked to 255 then conditional subtraction keeps the bias < 2^-1)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    raw = rng.integers(0, 256, size=(count, 32), dtype=np.uint8)
    raw[:, 31] &= 0x3F     # < 2^254 < r : uniform on [0, 2^254), plenty for a workload generator
    return raw.reshape(-1)


def random_r1cs(n, m, seed, nnz=(1, 16), one=True, unused=None, special=0.25):
    """A general R1CS family for parity: nothing of the benchmark circuits' regular shape.

      - every l, r and lhs row has between nnz[0] and nnz[1] entries (distinct, sorted columns) -- multi-term on BOTH operands
        (Gate {lhs; l; r} with arbitrary Affine.t, src/lib/zk/circuit.ml:73-75; QAP.ml:30-52 reads each coefficient off);
      - columns repeat freely across rows; `unused` variables (default max(1, m // 8)) occur in NO row, so their v_k, w_k, y_k are the
        zero polynomial and their key points the identity (groth16.ml:59-68,74-79 with L_k = 0);
      - coefficients: small integers, their negatives, and full-width field elements, mixed;
      - witness values 0, 1 and r - 1 mixed into uniformly random ones (`special` = their share);
      - one=False: no variable is pinned to 1 (the reference's `x * x` program has no $ONE, src/lib/test/test.ml:204-212);
      - satisfied by construction: the last entry of every lhs row is solved for.
    Variables 0..m-1 in Var.compare order as everywhere; roughly three quarters are mids (all but 64 from 256 variables up), ONE (variable 0, when present) is public.
    Returns (R1CS, witness ints).  Vectorised (numpy object arrays): 2^20 rows of 8 entries take seconds."""
    P = FR_MODULUS
    rng = np.random.Generator(np.random.PCG64(seed))
    lo, hi = nnz
    unused = max(1, m // 8) if unused is None else unused
    used = m - unused
    assert n >= 1 and 1 <= lo <= hi and used >= max(2, hi + 1)
    # which variables are in use: a random subset, ONE always
    perm = rng.permutation(m - 1) + 1 if one else rng.permutation(m)
    live = np.sort(np.concatenate(([0], perm[:used - 1])) if one else perm[:used]).astype(np.int64)
    # witness
    raw = rng.integers(0, 256, size=(m, 32), dtype=np.uint8)
    raw[:, 31] &= 0x3F
    w = [int.from_bytes(bytes(raw[k]), "little") for k in range(m)]
    kind = rng.random(m)
    for k in range(m):
        if kind[k] < special:
            w[k] = (0, 1, P - 1)[int(kind[k] * 3 / special) % 3]
    if one:
        w[0] = 1
    nz = [int(k) for k in live if w[k] != 0]
    if len(nz) < 4:                                         # the solved-for entry needs a non-zero witness value
        for k in live[:4]:
            if w[k] == 0:
                w[k] = 2 + int(k)
        nz = [int(k) for k in live if w[k] != 0]
    pivots = np.array(nz[:64], dtype=np.int64)
    pivot_inv = np.array([pow(w[k], P - 2, P) for k in pivots], dtype=object)
    wobj = np.array(w, dtype=object)

    def rows(count_lo, count_hi, exclude=None):
        cnt = rng.integers(count_lo, count_hi + 1, size=n)
        ptr = np.zeros(n + 1, dtype=np.int64)
        ptr[1:] = np.cumsum(cnt)
        tot = int(ptr[-1])
        row = np.repeat(np.arange(n, dtype=np.int64), cnt)
        first = ptr[:-1][row]
        # distinct columns per row: a random start plus strictly increasing steps whose total stays below `span`
        span = used if exclude is None else used - 1
        step = rng.integers(1, max(2, (span - 1) // hi + 1), size=tot)
        off = np.cumsum(step) - np.cumsum(step)[first] + step[first]
        start = rng.integers(0, span, size=n)[row]
        pos = (start + off) % span
        if exclude is not None:                             # positions index the live variables without the row's pivot
            pos = pos + (pos >= exclude[row])
        col = live[pos]
        order = np.lexsort((col, row))
        col = col[order]
        ck = rng.integers(0, 8, size=tot)
        small = rng.integers(1, 1 << 31, size=tot).astype(object)
        big = rng.integers(0, 256, size=(tot, 32), dtype=np.uint8)
        big[:, 31] &= 0x3F
        bigv = np.array([int.from_bytes(bytes(b), "little") for b in big[ck >= 6]], dtype=object) if (ck >= 6).any() else np.array([], dtype=object)
        coef = small.copy()
        coef[ck == 0] = 1
        coef[ck == 1] = P - 1
        neg = (ck == 2) | (ck == 3)
        coef[neg] = P - small[neg]
        coef[ck >= 6] = bigv
        return ptr, row, col, coef

    def dots(ptr, col, coef):
        prod = coef * wobj[col]
        return np.add.reduceat(prod, ptr[:-1]) % P

    lp, _, lc, lv = rows(lo, hi)
    rp, _, rc, rv = rows(lo, hi)
    a, b = dots(lp, lc, lv), dots(rp, rc, rv)
    want = a * b % P
    piv_i = rng.integers(0, len(pivots), size=n)
    piv = pivots[piv_i]
    piv_pos = np.searchsorted(live, piv)
    if hi > 1:
        op, orow, oc, ov = rows(max(lo - 1, 0) if lo > 1 else 0, hi - 1, exclude=piv_pos)
        part = np.zeros(n, dtype=object)
        nonempty = op[1:] > op[:-1]
        if nonempty.any():
            red = np.add.reduceat(ov * wobj[oc], op[:-1][nonempty]) % P
            part[nonempty] = red
    else:
        op, orow, oc, ov = np.zeros(n + 1, dtype=np.int64), np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64), np.zeros(0, dtype=object)
        part = np.zeros(n, dtype=object)
    solved = (want - part) % P * pivot_inv[piv_i] % P
    # merge the solved-for entry into each lhs row at its sorted place
    cnt = (op[1:] - op[:-1]) + 1
    optr = np.zeros(n + 1, dtype=np.int64)
    optr[1:] = np.cumsum(cnt)
    allrow = np.concatenate((orow, np.arange(n, dtype=np.int64)))
    allcol = np.concatenate((oc, piv))
    allval = np.concatenate((ov, solved))
    order = np.lexsort((allcol, allrow))
    ocol, oval = allcol[order], allval[order]

    def matrix(ptr, col, coef):
        vals = np.frombuffer(b"".join(int(c).to_bytes(32, "little") for c in coef), dtype=np.uint8).copy() if len(coef) else np.zeros(0, dtype=np.uint8)
        return Matrix(ptr.astype(np.uint32), col.astype(np.uint32), vals)

    mid = (rng.random(m) < 0.75).astype(np.uint8)
    if m > 256:                                             # a statement has a handful of public values, not a quarter of its variables: at most 64
        pub = np.flatnonzero(mid == 0)
        mid[pub[64:]] = 1
    if one:
        mid[0] = 0
    if mid.all():
        mid[m - 1] = 0
    return R1CS(n, m, matrix(lp, lc, lv), matrix(rp, rc, rv), matrix(optr, ocol, oval), mid), w
