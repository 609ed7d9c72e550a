"""CPU-only checks of the drop-in boundary: libzkmi355x.so loads, exports every symbol that
include/zkmi355x.h declares, fails loudly without a GPU (no CPU fallback), and its pure byte-level
helpers agree with the oracle."""
import ctypes as C
import os
import re

import pytest

import oracle_lib as O
from oracle import pyref as P
from zukelang_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "zkmi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), "missing export: " + s
    assert sorted(_lib.EXPORTS) == syms, "zukelang_amd/_lib.py EXPORTS is out of sync with include/zkmi355x.h"


def test_no_torch_types_in_the_header():
    text = open(os.path.join(ROOT, "include", "zkmi355x.h")).read()
    assert 'extern "C"' in text
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    assert "torch" not in code and "at::" not in code and "std::" not in code and "template" not in code


def test_strerror_codes():
    lib = _lib.lib()
    assert lib.zk_strerror(0) == b"ok"
    for code in range(-8, 0):
        assert lib.zk_strerror(code) not in (b"ok", b"unknown error")
    assert lib.zk_strerror(-99) == b"unknown error"


def _gpu_present():
    return _lib.lib().zk_device_count() > 0


@pytest.mark.skipif(_gpu_present(), reason="a GPU is visible")
def test_fails_loudly_without_a_gpu():
    """The product path has no CPU fallback: every compute entry point reports ZK_ERR_HIP."""
    lib = _lib.lib()
    assert lib.zk_device_count() == 0
    assert lib.zk_init(0) == -5
    assert b"no HIP device" in lib.zk_last_error()
    # the device-list entry points (multi-device keys) fail the same way and leave no list behind
    assert lib.zk_set_devices(C.c_uint64(0b11)) == -5 and b"no HIP device" in lib.zk_last_error()
    devs = (C.c_int32 * 2)(0, 0)
    assert lib.zk_set_device_list(devs, C.c_uint32(2)) == -5
    assert lib.zk_set_devices(C.c_uint64(0)) == -1 and lib.zk_set_device_list(None, C.c_uint32(0)) == -1          # argument errors come first
    cnt = C.c_uint32(7)
    assert lib.zk_get_device_list(None, C.c_uint32(0), C.byref(cnt)) == 0 and cnt.value == 0
    buf = (C.c_uint8 * 64)()
    assert lib.zk_fr_ntt(buf, 1, 0) == -5
    out = (C.c_uint8 * 96)()
    g = O.g1_generator()
    assert lib.zk_msm_g1(g, C.c_size_t(1), P.fr_to_bytes(1), C.c_size_t(1), 0, out) == -5
    from zukelang_amd.curve import FFT_Fr
    with pytest.raises(_lib.ZkError) as e:
        FFT_Fr.fft(P.fr_to_bytes(1) * 2, 1)
    assert e.value.code == -5


def test_argument_errors_need_no_gpu():
    lib = _lib.lib()
    out = (C.c_uint8 * 96)()
    g = O.g1_generator()
    # curve.ml:116 invalid_arg "apply_powers": more coefficients than points
    assert lib.zk_msm_g1(g, C.c_size_t(1), P.fr_to_bytes(1) * 2, C.c_size_t(2), 0, out) == -6
    assert lib.zk_msm_g1(g, C.c_size_t(1), P.fr_to_bytes(1), C.c_size_t(1), 0, None) == -1
    assert lib.zk_fr_ntt(None, 3, 0) == -1
    assert lib.zk_groth16_pk_free(C.c_uint64(12345)) == -7
    n = C.c_size_t(99)
    assert lib.zk_fr_poly_mul(None, C.c_size_t(0), None, C.c_size_t(0), None, C.byref(n)) == 0 and n.value == 0


def test_compress_matches_oracle():
    """to_compressed_bytes (curve.ml:199,208) is pure byte logic in the library."""
    lib = _lib.lib()
    for k in (1, 2, 3, 5, 12345, P.R - 1, P.R - 2, 0x1234567890ABCDEF):
        p1 = O.g1_mul(O.g1_generator(), P.fr_to_bytes(k))
        p2 = O.g2_mul(O.g2_generator(), P.fr_to_bytes(k))
        o1 = (C.c_uint8 * 48)()
        o2 = (C.c_uint8 * 96)()
        assert lib.zk_g1_compress(p1, o1) == 0 and bytes(o1) == O.g1_compress(p1)
        assert lib.zk_g2_compress(p2, o2) == 0 and bytes(o2) == O.g2_compress(p2)
    inf1 = bytes([0x40]) + bytes(95)
    o1 = (C.c_uint8 * 48)()
    assert lib.zk_g1_compress(inf1, o1) == 0 and bytes(o1) == bytes([0xC0]) + bytes(47)


def test_shard_range_is_the_rule_the_host_side_mirrors():
    """zk_groth16_shard_range (pure host arithmetic in the library) == groth16.shard_bounds, uniform and with a heavy prefix."""
    from zukelang_amd.groth16 import shard_bounds
    lib = _lib.lib()
    lo, hi = C.c_uint64(), C.c_uint64()
    for size in (13, 65545, 262147, (1 << 24) + 3):
        for heavy in (0, 5, size // 4 + 3, size):
            for world in (1, 2, 3, 8):
                for rank in range(world):
                    assert lib.zk_groth16_shard_range(C.c_uint64(size), C.c_uint64(heavy), rank, world, C.byref(lo), C.byref(hi)) == 0
                    assert (lo.value, hi.value) == shard_bounds(size, rank, world, heavy)
    assert lib.zk_groth16_shard_range(C.c_uint64(10), C.c_uint64(0), 2, 2, C.byref(lo), C.byref(hi)) == -1


def test_a_plain_c99_host_binds_the_abi(tmp_path):
    """include/zkmi355x.h is C (no C++-isms, no torch types): examples/c_host.c compiles with gcc -std=c99 -pedantic, links against the
    library and runs its GPU-free calls."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_host")
    libdir = os.path.join(root, "zukelang_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "examples", "c_host.c"), "-o", exe, "-L" + libdir, "-lzkmi355x", "-Wl,-rpath," + libdir])
    out = subprocess.check_output([exe])
    assert out.startswith(b"c-host ok")


def test_set_option_accepts_the_public_knobs_and_nothing_else():
    """zk_set_option needs no GPU: names with or without the ZK_ prefix, in either case; NULL hands the knob back to the environment."""
    L = _lib.lib()
    assert L.zk_set_option(b"no_such_knob", b"1") == -1 and L.zk_set_option(None, b"1") == -1 and L.zk_set_option(b"", b"1") == -1
    assert L.zk_set_option(b"test_forms", b"1") == -1                  # test-only switches stay environment-only
    for name in (b"msm_window", b"ZK_MSM_WINDOW", b"Msm_Window", b"key_subgroup_check", b"slot_streams", b"graph", b"derive_side_by_side", b"pin_compact_h", b"PIN_SHARED_SORT"):
        assert L.zk_set_option(name, b"16") == 0
        assert L.zk_set_option(name, None) == 0
