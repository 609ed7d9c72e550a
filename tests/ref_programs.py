"""Loader of tests/golden/ref_programs.json (the reference's own acceptance programs, src/lib/test/test.ml:194-276, as R1CS rows,
witnesses and first-principles proofs; written by tests/golden/make_ref_programs.py) -- test infrastructure."""
import json
import os

import numpy as np

from zukelang_amd import r1cs as RC

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_programs.json")
_DOC = None


def doc():
    global _DOC
    if _DOC is None:
        _DOC = json.load(open(_PATH))
    return _DOC


def names():
    return [p["name"] for p in doc()["programs"]]


def program(name):
    return next(p for p in doc()["programs"] if p["name"] == name)


def circuit(p):
    """The R1CS of one program (explicit zero coefficients kept, rows with no entry kept)."""
    rows = lambda M: [{int(k): int(c) for k, c in row} for row in M]
    return RC.R1CS(p["n"], p["m"], RC.Matrix.from_rows(rows(p["L"])), RC.Matrix.from_rows(rows(p["R"])), RC.Matrix.from_rows(rows(p["O"])),
                   np.array(p["mid"], dtype=np.uint8))


def witnesses(p):
    return [[int(x) for x in w["sol"]] for w in p["witnesses"]]


def groth16_params(p):
    g = p["groth16"]
    return [int(x) for x in g["toxic_alpha_beta_gamma_delta_tau"]], int(g["r"]), int(g["s"])


def pinocchio_params(p):
    g = p["pinocchio"]
    return [int(x) for x in g["toxic_rv_rw_s_av_aw_ay_b_gm"]], (int(g["dv"]), int(g["dw"]), int(g["dy"]))


def groth16_proof(w):
    g = w["groth16_proof"]
    return bytes.fromhex(g["a"]), bytes.fromhex(g["b"]), bytes.fromhex(g["c"])


_PORDER = ("vv", "ww", "yy", "h", "vavv", "waww", "yayy", "bvwy")           # Compute.proof, pinocchio.ml:195-208


def pinocchio_proof(w, zk=True):
    g = w["pinocchio_zk_proof" if zk else "pinocchio_nonzk_proof"]
    return b"".join(bytes.fromhex(g[k]) for k in _PORDER)
