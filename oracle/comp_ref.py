"""CPU restatement of the reference's DSL front end -- TEST INFRASTRUCTURE ONLY.

What it is for: the reference's own acceptance programs (src/lib/test/test.ml:194-276) have to enter
the prove path as R1CS rows, and there is no OCaml here to run Comp.compile.  This module restates, in
Python integers, exactly the parts of the reference that decide what those rows are:

  Lang.Expr.Combinator      src/lib/zk/lang.ml:160-247     (which Var.make calls a program makes)
  Lang.Eval.eval            src/lib/zk/lang.ml:319-427     (the interpreter the harness compares with)
  Comp.compile              src/lib/zk/comp.ml:193-444     (DSL -> gates + straight-line witness code)
  Comp.fix_output           src/lib/zk/comp.ml:448-473
  Comp.compile (final)      src/lib/zk/comp.ml:491-530     (inputs_public / outputs / mids)
  Comp.Code.eval / eval_list src/lib/zk/comp.ml:71-122
  Comp.compile_value        src/lib/zk/comp.ml:130-146
  Circuit.Affine            src/lib/zk/circuit.ml:8-71     (sparse linear forms; `add` KEEPS zero coefficients)
  Circuit.Gate.compare      src/lib/zk/circuit.ml:85-91    (gate order = QAP point order, QAP.ml:22)
  Var.make / Var.compare    src/lib/zk/var.ml:8-18         (global counter; polymorphic compare on (string * int))

Only tests/ (and the fixture generator under tests/golden/) import it.  Nothing here was produced by
running the reference: it is read from the OCaml text.

One thing the text cannot settle: `Gate.compare` falls back to `F.compare` when two affine forms have the same
variable with different coefficients, and `Bls12_381.Fr.compare` lives in the external opam package
(bls12-381 6.1.0, not under /root/reference).  `fr_compare` below is therefore a PARAMETER: "numeric"
(compare of to_z) or "bytes_le" (compare of the 32-byte little-endian to_bytes).  A program whose gate order
depends on it reports `order_depends_on_fr_compare = True`.
"""
from dataclasses import dataclass, field as dc_field

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
OMEGA_2_32 = pow(5, (R - 1) >> 32, R)          # Root_of_unity: curve.ml:241-298 (first g with a primitive root is 5; FFT.ml:208)

ONE = ("ONE", 1)                               # circuit.ml:3: created at module initialisation, counter value 1


class DivisionByZero(Exception):
    """F.(a / b) with b = 0 (comp.ml:88-91; the harness retries with other inputs, test.ml:148-150)."""


def f_of_int(i):
    return i % R


def f_of_uint32(i):
    """lang.ml:7-10 -> Root_of_unity.f_of_uint 32 i = g ** (i lsl (32 - 32)), g the primitive 2^32-th root."""
    return pow(OMEGA_2_32, i, R)


# ------------------------------------------------------------------ Var (var.ml)
class VarGen:
    """Var.make (var.ml:14-18): one global counter.  It starts at 1 because Circuit.one took that value."""

    def __init__(self, start=1):
        self.cntr = start

    def make(self, prefix):
        self.cntr += 1
        return (prefix, self.cntr)


# ------------------------------------------------------------------ Affine (circuit.ml:8-71)
def aff_add(a, b):
    out = dict(a)
    for v, f in b.items():
        out[v] = (out[v] + f) % R if v in out else f          # union (fun _ f1 f2 -> Some (f1 + f2)): zeros stay
    return out


def aff_scale(a, f):
    return {v: c * f % R for v, c in a.items()}


def aff_of_F(f):
    return {} if f % R == 0 else {ONE: f % R}


def aff_of_int(i):
    return aff_of_F(f_of_int(i))


def aff_sub(a, b):
    return a if not b else aff_add(a, aff_scale(b, R - 1))


def aff_is_const(a):
    rest = {v: c for v, c in a.items() if v != ONE}
    if not rest:
        return a.get(ONE, 0)
    return None


def aff_bindings(a):
    return sorted(a.items())                                    # Var.Map.bindings: Var.compare order


def aff_eval(env, a):
    return sum(env[v] * c for v, c in a.items()) % R


# ------------------------------------------------------------------ Lang types / expressions (lang.ml:28-252)
FIELD, BOOL, UINT32 = ("field",), ("bool",), ("uint32",)


def ty_pair(a, b):
    return ("pair", a, b)


def ty_either(a, b):
    return ("either", a, b)


@dataclass
class Expr:
    desc: tuple
    ty: tuple


class Lang:
    """Lang.Expr.Combinator.  A program is built through one instance so that its Var.make calls hit the shared counter."""

    def __init__(self, vargen):
        self.vg = vargen

    def bool_(self, b): return Expr(("Bool", bool(b)), BOOL)
    def field(self, n): return Expr(("Field", n % R), FIELD)
    def uint32(self, n): return Expr(("Uint32", n), UINT32)
    def const(self, n): return Expr(("Field", f_of_int(n)), FIELD)            # (!)
    def add(self, a, b): return Expr(("Add", a, b), FIELD)
    def sub(self, a, b): return Expr(("Sub", a, b), FIELD)
    def neg(self, a): return Expr(("Neg", a), FIELD)
    def mul(self, a, b): return Expr(("Mul", a, b), FIELD)
    def div(self, a, b): return Expr(("Div", a, b), FIELD)
    def not_(self, a): return Expr(("Not", a), BOOL)
    def and_(self, a, b): return Expr(("And", a, b), BOOL)
    def or_(self, a, b): return Expr(("Or", a, b), BOOL)
    def if_(self, a, b, c): return Expr(("If", a, b, c), b.ty)
    def input(self, name, sec, ty): return Expr(("Input", name, sec), ty)
    def eq(self, a, b): return Expr(("Eq", a, b), BOOL)
    def pair(self, a, b): return Expr(("Pair", a, b), ty_pair(a.ty, b.ty))
    def left(self, a, bty): return Expr(("Left", a), ty_either(a.ty, bty))
    def right(self, aty, b): return Expr(("Right", b), ty_either(aty, b.ty))
    def add_u32(self, a, b): return Expr(("Add_uint32", a, b), UINT32)
    def sub_u32(self, a, b): return Expr(("Sub_uint32", a, b), UINT32)

    def to_field(self, t):
        assert t.ty in (FIELD, BOOL, UINT32)
        return Expr(("To_field", t), FIELD)

    def fst(self, a):
        assert a.ty[0] == "pair"
        return Expr(("Fst", a), a.ty[1])

    def snd(self, a):
        assert a.ty[0] == "pair"
        return Expr(("Snd", a), a.ty[2])

    def let_(self, a, body):                                     # lang.ml:217-220: Var.make "x" BEFORE the body is built
        v = self.vg.make("x")
        b = body(Expr(("Var", v), a.ty))
        return Expr(("Let", v, a, b), b.ty)

    def case(self, ab, fa, fb):                                  # lang.ml:236-245: va, then vb, then the branches
        assert ab.ty[0] == "either"
        va = self.vg.make("case")
        vb = self.vg.make("case")
        a = fa(Expr(("Var", va), ab.ty[1]))
        b = fb(Expr(("Var", vb), ab.ty[2]))
        return Expr(("Case", ab, va, a, vb, b), a.ty)


# ------------------------------------------------------------------ Lang.Eval (lang.ml:319-427); values: ("Field", f) ("Bool", b) ("Uint32", i) ("Pair", a, b) ("Left", a) ("Right", b)
def lang_eval(inputs, e, env=None):
    env = env or {}
    d = e.desc
    k = d[0]
    ev = lambda x, en=env: lang_eval(inputs, x, en)
    if k == "Input": return inputs[d[1]]
    if k == "Field": return ("Field", d[1])
    if k == "Uint32": return ("Uint32", d[1])
    if k == "Bool": return ("Bool", d[1])
    if k == "Add": return ("Field", (ev(d[1])[1] + ev(d[2])[1]) % R)
    if k == "Sub": return ("Field", (ev(d[1])[1] - ev(d[2])[1]) % R)
    if k == "Mul": return ("Field", ev(d[1])[1] * ev(d[2])[1] % R)
    if k == "Div":
        b = ev(d[2])[1]
        if b == 0: raise DivisionByZero()
        return ("Field", ev(d[1])[1] * pow(b, R - 2, R) % R)
    if k == "Not": return ("Bool", not ev(d[1])[1])
    if k == "And": return ("Bool", ev(d[1])[1] and ev(d[2])[1])
    if k == "Or": return ("Bool", ev(d[1])[1] or ev(d[2])[1])
    if k == "If": return ev(d[2]) if ev(d[1])[1] else ev(d[3])
    if k == "Eq": return ("Bool", ev(d[1]) == ev(d[2]))
    if k == "To_field":
        v = ev(d[1])
        if v[0] == "Field": return v
        if v[0] == "Bool": return ("Field", 1 if v[1] else 0)
        if v[0] == "Uint32": return ("Field", f_of_uint32(v[1]))
        raise AssertionError
    if k == "Let":
        en = dict(env); en[d[1]] = ev(d[2])
        return lang_eval(inputs, d[3], en)
    if k == "Var": return env[d[1]]
    if k == "Neg": return ("Field", (R - ev(d[1])[1]) % R)
    if k == "Pair": return ("Pair", ev(d[1]), ev(d[2]))
    if k == "Fst": return ev(d[1])[1]
    if k == "Snd": return ev(d[1])[2]
    if k == "Left": return ("Left", ev(d[1]))
    if k == "Right": return ("Right", ev(d[1]))
    if k == "Case":
        v = ev(d[1])
        en = dict(env)
        if v[0] == "Left":
            en[d[2]] = v[1]; return lang_eval(inputs, d[3], en)
        en[d[4]] = v[1]; return lang_eval(inputs, d[5], en)
    if k == "Add_uint32":
        c = ev(d[1])[1] + ev(d[2])[1]
        return ("Uint32", c - (1 << 32) if c >= 1 << 32 else c)
    if k == "Sub_uint32":
        c = ev(d[1])[1] - ev(d[2])[1]
        return ("Uint32", c + (1 << 32) if c < 0 else c)
    raise AssertionError(k)


# ------------------------------------------------------------------ Comp (comp.ml)
def components(ty):                                              # comp.ml:125-128
    if ty[0] in ("field", "bool", "uint32"): return 1
    if ty[0] == "pair": return components(ty[1]) + components(ty[2])
    return max(components(ty[1]), components(ty[2])) + 1


def compile_value(ty, v):                                        # comp.ml:130-146
    if v[0] == "Field": return [v[1]]
    if v[0] == "Bool": return [1 if v[1] else 0]
    if v[0] == "Uint32": return [f_of_uint32(v[1])]
    if v[0] == "Pair": return compile_value(ty[1], v[1]) + compile_value(ty[2], v[2])
    cs = components(ty) - 1
    if v[0] == "Left":
        fs = compile_value(ty[1], v[1]); return [0] + fs + [0] * (cs - len(fs))
    fs = compile_value(ty[2], v[1]); return [1] + fs + [0] * (cs - len(fs))


# Code.t (comp.ml:22-29): ("Mul", a, b) ("Div", a, b) ("Not", a) ("Or", a, b) ("Affine", aff) ("Eq", a, b) ("If", a, b, c)
def code_eval(env, c):                                           # comp.ml:71-112
    k = c[0]
    def to_bool(f):
        assert f in (0, 1)
        return f == 1
    if k == "Mul": return code_eval(env, c[1]) * code_eval(env, c[2]) % R
    if k == "Div":
        a, b = code_eval(env, c[1]), code_eval(env, c[2])
        if b == 0: raise DivisionByZero()
        return a * pow(b, R - 2, R) % R
    if k == "Not": return 0 if to_bool(code_eval(env, c[1])) else 1
    if k == "Or":
        a, b = code_eval(env, c[1]), code_eval(env, c[2])
        return 1 if (to_bool(a) or to_bool(b)) else 0
    if k == "Eq": return 1 if code_eval(env, c[1]) == code_eval(env, c[2]) else 0
    if k == "If": return code_eval(env, c[2]) if to_bool(code_eval(env, c[1])) else code_eval(env, c[3])
    if k == "Affine": return aff_eval(env, c[1])
    raise AssertionError(k)


def code_eval_list(env, codes):                                  # comp.ml:114-122
    env = dict(env)
    for v, c in codes:
        assert v not in env
        env[v] = code_eval(env, c)
    return env


def _fr_cmp(kind):
    if kind == "numeric":
        return lambda a, b: (a > b) - (a < b)
    if kind == "bytes_le":
        return lambda a, b: (a.to_bytes(32, "little") > b.to_bytes(32, "little")) - (a.to_bytes(32, "little") < b.to_bytes(32, "little"))
    raise ValueError(kind)


class _Order:
    """Var.Map.compare F.compare on two affine forms (circuit.ml:35), recording whether F.compare ever decided."""

    def __init__(self, kind):
        self.cmp = _fr_cmp(kind)
        self.used_fr = False

    def affine(self, a, b):
        la, lb = aff_bindings(a), aff_bindings(b)
        for (va, fa), (vb, fb) in zip(la, lb):
            if va != vb: return -1 if va < vb else 1
            c = self.cmp(fa, fb)
            if c:
                self.used_fr = True
                return c
        return (len(la) > len(lb)) - (len(la) < len(lb))

    def gate(self, g, h):                                        # circuit.ml:85-91: lhs, then l, then r
        for x, y in zip(g, h):
            c = self.affine(x, y)
            if c: return c
        return 0


@dataclass
class Compiled:
    gates: list                 # [(lhs, l, r)] as inserted (a Gate.Set: duplicates dropped)
    inputs: dict                # name -> (security, ty, [vars]); "$ONE" included when add_one ran
    codes: list                 # [(var, code)] in execution order
    result: list                # the output affines after fix_output
    inputs_public: set = dc_field(default_factory=set)
    outputs: set = dc_field(default_factory=set)
    mids: set = dc_field(default_factory=set)


class Comp:
    """Comp.Make(F).compile.  State = GateM.state (comp.ml:148-192)."""

    def __init__(self, vargen):
        self.vg = vargen
        self.gates = []
        self.inputs = {}
        self.rev_codes = []

    # GateM
    def add_gate(self, lhs, l, r):
        g = (dict(lhs), dict(l), dict(r))
        if g not in self.gates:                                   # Gate.Set.add
            self.gates.append(g)

    def add_one(self):
        if "$ONE" not in self.inputs:
            self.inputs["$ONE"] = ("public", FIELD, [ONE])

    def add_input(self, name, sec, ty):
        assert name != "$ONE"
        if name in self.inputs: raise ValueError("duplicated input name")
        vs = [self.vg.make(name) for _ in range(components(ty))]
        self.inputs[name] = (sec, ty, vs)
        return [{v: 1} for v in vs]

    def add_code(self, v, code):
        assert all(v != w for w, _ in self.rev_codes)
        self.rev_codes.append((v, code))

    def var(self):
        v = self.vg.make("c")
        return v, {v: 1}

    def compile1(self, env, e):
        res = self.compile(env, e)
        assert len(res) == 1
        return res[0]

    def compile(self, env, e):                                    # comp.ml:202-444
        d = e.desc
        k = d[0]
        A = lambda a: ("Affine", a)
        one_ = aff_of_int(1)
        zero_ = aff_of_int(0)
        if k == "Field":
            self.add_one(); return [aff_of_F(d[1])]
        if k == "Bool": return [one_ if d[1] else zero_]
        if k == "Uint32": return [aff_of_F(f_of_uint32(d[1]))]
        if k == "Input": return self.add_input(d[1], d[2], e.ty)
        if k == "Add":
            t1 = self.compile1(env, d[1]); t2 = self.compile1(env, d[2]); return [aff_add(t1, t2)]
        if k == "Sub":
            return self.compile(env, Expr(("Add", d[1], Expr(("Neg", d[2]), FIELD)), FIELD))
        if k == "Neg":
            return [aff_scale(self.compile1(env, d[1]), R - 1)]
        if k == "Mul":
            t1 = self.compile1(env, d[1]); t2 = self.compile1(env, d[2])
            f1, f2 = aff_is_const(t1), aff_is_const(t2)
            if f1 is not None and f2 is not None: return [aff_of_F(f1 * f2 % R)]
            if f1 is not None: return [aff_scale(t2, f1)]
            if f2 is not None: return [aff_scale(t1, f2)]
            va, a = self.var()
            self.add_code(va, ("Mul", A(t1), A(t2)))
            self.add_gate(a, t1, t2)
            return [a]
        if k == "Div":
            a = self.compile1(env, d[1]); b = self.compile1(env, d[2])
            fa, fb = aff_is_const(a), aff_is_const(b)
            if fa is not None and fb is not None: return [aff_of_F(fa * fb % R)]      # sic (comp.ml:249): the reference multiplies
            if fa is not None: return [aff_scale(b, fa)]                               # sic (comp.ml:250)
            if fb is not None: return [aff_scale(a, fb)]                               # sic (comp.ml:251)
            vc, c = self.var(); vd, dd = self.var()
            self.add_code(vc, ("Div", A(one_), A(b)))
            self.add_code(vd, ("Mul", A(a), A(c)))
            self.add_one()
            self.add_gate(one_, b, c)
            self.add_gate(dd, a, c)
            return [dd]
        if k == "Not":
            if d[1].desc[0] == "Bool":
                return self.compile(env, Expr(("Bool", not d[1].desc[1]), BOOL))
            a = self.compile1(env, d[1])
            vb, b = self.var()
            self.add_code(vb, ("Not", A(a)))
            self.add_one()
            self.add_gate(zero_, a, b)
            self.add_gate(one_, aff_add(a, b), one_)
            return [b]
        if k == "And":
            return self.compile(env, Expr(("Mul", Expr(("To_field", d[1]), FIELD), Expr(("To_field", d[2]), FIELD)), FIELD))
        if k == "Or":
            a = self.compile1(env, d[1]); b = self.compile1(env, d[2])
            vc, c = self.var(); vd, dd = self.var()
            apb = aff_add(a, b)
            self.add_one()
            self.add_code(vc, ("Or", A(a), A(b)))
            self.add_code(vd, ("If", A(c), ("Div", A(one_), A(apb)), A(zero_)))
            self.add_gate(c, apb, dd)
            self.add_gate(zero_, apb, aff_sub(one_, c))
            return [c]
        if k == "If":
            a = self.compile1(env, d[1])
            fa = aff_is_const(a)
            if fa is not None:
                return self.compile(env, d[2] if fa == 1 else d[3])
            b = self.compile(env, d[2]); c = self.compile(env, d[3])
            out = []
            for bi, ci in zip(b, c):
                vd, dd = self.var()                               # allocated even when it stays unused (comp.ml:315)
                b_c = aff_sub(bi, ci)
                f = aff_is_const(b_c)
                if f is not None:
                    out.append(aff_add(ci, aff_scale(a, f)))
                else:
                    self.add_code(vd, ("Mul", A(a), A(b_c)))
                    self.add_gate(dd, a, b_c)
                    out.append(aff_add(ci, dd))
            return out
        if k == "Eq":
            a = self.compile(env, d[1]); b = self.compile(env, d[2])
            cs = []
            for ai, bi in zip(a, b):                              # the [a],[b] case and the general case emit the same per-component gates
                vc, c = self.var(); vd, dd = self.var()
                self.add_one()
                amb = aff_sub(ai, bi)
                self.add_code(vc, ("Eq", A(ai), A(bi)))
                self.add_code(vd, ("If", A(c), A(zero_), ("Div", A(one_), A(amb))))
                self.add_gate(aff_sub(one_, c), amb, dd)
                self.add_gate(zero_, amb, c)
                cs.append(c)
            acc = cs[0]
            for c in cs[1:]:
                vx, x = self.var()
                self.add_code(vx, ("Mul", A(acc), A(c)))
                self.add_gate(x, acc, c)
                acc = x
            return [acc]
        if k == "To_field": return self.compile(env, d[1])
        if k == "Let":
            a = self.compile(env, d[2])
            return self.compile([(d[1], a)] + env, d[3])
        if k == "Var":
            return next(a for v, a in env if v == d[1])          # List.assoc: first match
        if k == "Pair":
            return self.compile(env, d[1]) + self.compile(env, d[2])
        if k == "Fst":
            cs = components(d[1].ty[1]); return self.compile(env, d[1])[:cs]
        if k == "Snd":
            cs = components(d[1].ty[1]); return self.compile(env, d[1])[cs:]
        if k == "Left":
            return [zero_] + self.compile(env, d[1])
        if k == "Right":
            a = self.compile(env, d[1]); self.add_one(); return [one_] + a
        if k == "Case":
            ab_e, va, ce, vb, de = d[1], d[2], d[3], d[4], d[5]
            aty, bty = ab_e.ty[1], ab_e.ty[2]
            ab = self.compile(env, ab_e)
            tag = ab[0]
            for_a = ab[:components(aty) + 1][1:]
            for_b = ab[:components(bty) + 1][1:]
            c = self.compile([(va, for_a)] + env, ce)
            dd = self.compile([(vb, for_b)] + env, de)
            self.add_one()
            out = []
            for ci, di in zip(c, dd):
                vx, x = self.var(); vy, y = self.var()
                tm1 = aff_sub(tag, one_)
                self.add_code(vx, ("Mul", A(tm1), A(ci)))
                self.add_gate(x, tm1, ci)
                self.add_code(vy, ("Mul", A(tag), A(di)))
                self.add_gate(y, tag, di)
                out.append(aff_add(x, y))
            return out
        if k == "Add_uint32":
            return self.compile(env, Expr(("Mul", Expr(("To_field", d[1]), FIELD), Expr(("To_field", d[2]), FIELD)), FIELD))
        if k == "Sub_uint32":
            return self.compile(env, Expr(("Div", Expr(("To_field", d[1]), FIELD), Expr(("To_field", d[2]), FIELD)), FIELD))
        raise AssertionError(k)

    def fix_output(self, a):                                      # comp.ml:448-473
        b = aff_bindings(a)
        if not b: return a
        if len(b) == 1 and b[0][0] == ONE: return a
        if len(b) == 1 and b[0][1] == 1: return a
        vo = self.vg.make("v")
        o = {vo: 1}
        self.add_code(vo, ("Affine", a))
        self.add_one()
        self.add_gate(o, a, aff_of_int(1))
        return o


def gate_vars(gates):
    vs = set()
    for g in gates:
        for a in g: vs |= set(a)
    return vs


def compile_program(e, vargen):
    """Comp.compile (comp.ml:491-530)."""
    c = Comp(vargen)
    result = [c.fix_output(a) for a in c.compile([], e)]
    vars_ = gate_vars(c.gates)
    inputs_vars = {}
    for name in sorted(c.inputs):                                 # String.Map.fold: key order (irrelevant for a map result)
        sec, _ty, vs = c.inputs[name]
        for v in vs:
            if v in vars_: inputs_vars[v] = sec
    outputs = set()
    for a in result:
        b = aff_bindings(a)
        if len(b) == 1 and b[0][0] != ONE: outputs.add(b[0][0])
        elif len(b) == 0: pass
        else: raise AssertionError("output is neither a variable nor zero")          # comp.ml:513: `assert false`
    inputs_public = {v for v, s in inputs_vars.items() if s != "secret"}
    mids = (vars_ - (set(inputs_vars) | outputs)) | (vars_ - (inputs_public | outputs))
    out = Compiled(c.gates, c.inputs, list(c.rev_codes), result)
    out.inputs_public, out.outputs, out.mids = inputs_public, outputs, mids
    return out


def input_env(comp, values):
    """Comp.convert_inputs (comp.ml:550-567): input values -> F.t per flattened input variable; "$ONE" = 1."""
    env = {}
    for name, (_sec, ty, vs) in comp.inputs.items():
        v = ("Field", 1) if name == "$ONE" else values[name]
        for var, f in zip(vs, compile_value(ty, v)):
            env[var] = f
    return env


def r1cs_of(comp, fr_compare="numeric"):
    """What QAP.build reads off the gates (QAP.ml:18-52): gate ids in Gate.Set.elements order, one row of
    coefficients per gate and matrix, variables = Circuit.vars in Var.compare order.
    Returns dict(vars, rows_l, rows_r, rows_o (lists of {var_index: coeff}, explicit zeros kept), mid flags, order_depends_on_fr_compare)."""
    import functools
    order = _Order(fr_compare)
    gates = sorted(comp.gates, key=functools.cmp_to_key(order.gate))
    vars_ = sorted(gate_vars(comp.gates))
    idx = {v: i for i, v in enumerate(vars_)}
    rows = lambda sel: [{idx[v]: f for v, f in g[sel].items()} for g in gates]
    return {"vars": vars_, "L": rows(1), "R": rows(2), "O": rows(0), "mid": [1 if v in comp.mids else 0 for v in vars_],
            "order_depends_on_fr_compare": order.used_fr, "gates": gates}


def witness_of(comp, values):
    """The harness' `sol` (test.ml:125-151): inputs restricted to the circuit's variables, then Code.eval_list."""
    vars_ = gate_vars(comp.gates)
    env = {v: f for v, f in input_env(comp, values).items() if v in vars_}
    return code_eval_list(env, comp.codes)
