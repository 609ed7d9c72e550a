"""ctypes loader of libzkmi355x.so -- the HIP product library (include/zkmi355x.h).

There is no CPU fallback: if the shared object is missing, or no MI355X is visible when a
compute entry point is called, the error is raised to the caller.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZK_LIBZKMI355X_PATH: load another build of the SAME library (the host-only sanitizer build of `make asan-host`, tests/test_sanitizers.py).
# Not a fallback: a missing file or a missing entry point still fails loudly.
LIB_PATH = os.environ.get("ZK_LIBZKMI355X_PATH") or os.path.join(_HERE, "libzkmi355x.so")

# every symbol include/zkmi355x.h declares (checked by tests/test_abi.py without a GPU)
EXPORTS = [
    "zk_strerror", "zk_last_error", "zk_device_count", "zk_init", "zk_shutdown", "zk_set_devices", "zk_set_device_list", "zk_get_device_list", "zk_set_option",
    "zk_fr_ntt", "zk_fr_poly_mul", "zk_fr_spmv", "zk_msm_g1", "zk_msm_g2", "zk_g1_of_fr", "zk_g2_of_fr",
    "zk_g1_powers", "zk_g2_powers", "zk_g1_compress", "zk_g2_compress", "zk_g1_decompress", "zk_g2_decompress", "zk_g1_decompress_batch", "zk_g2_decompress_batch",
    "zk_groth16_pk_upload", "zk_groth16_pk_upload_lagrange", "zk_groth16_pk_derive_lagrange", "zk_groth16_lagrange_pool_sizes", "zk_groth16_pk_derive_lagrange_sets", "zk_groth16_pk_install_lagrange", "zk_groth16_pk_shard", "zk_groth16_shard_range", "zk_groth16_pool_points", "zk_groth16_pk_free", "zk_groth16_prove", "zk_groth16_reserve_slots", "zk_groth16_prove_async", "zk_groth16_prove_wait", "zk_groth16_set_witness", "zk_groth16_qap_eval",
    "zk_groth16_pk_upload_sharded", "zk_groth16_prove_partial", "zk_groth16_prove_partial_async", "zk_groth16_prove_partial_wait", "zk_groth16_combine", "zk_groth16_prove_partial_wait_device", "zk_groth16_combine_device",
    "zk_groth16_pool_layout", "zk_groth16_scalars_async", "zk_groth16_scalars_wait", "zk_groth16_msm_partial_async",
    "zk_device_malloc", "zk_device_free", "zk_device_memcpy",
    "zk_pinocchio_pk_upload", "zk_pinocchio_pk_derive_lagrange", "zk_pinocchio_pool_points", "zk_pinocchio_pk_free", "zk_pinocchio_prove",
    "zk_pinocchio_reserve_slots", "zk_pinocchio_set_witness", "zk_pinocchio_prove_async", "zk_pinocchio_prove_wait",
    "zk_pairing_product", "zk_pairing_check", "zk_groth16_verify", "zk_pinocchio_verify",
    "zk_profile_enable", "zk_profile_reset", "zk_profile_get", "zk_profile_names", "zk_profile_counter", "zk_sync",
    "zk_bench_field_mul", "zk_selftest_fp",
]


class ZkError(RuntimeError):
    def __init__(self, code, detail):
        self.code = code
        super().__init__("libzkmi355x error %d: %s" % (code, detail))


class CSR(C.Structure):
    _fields_ = [("row_ptr", C.POINTER(C.c_uint32)), ("col", C.POINTER(C.c_uint32)), ("val", C.POINTER(C.c_uint8))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.zk_strerror.restype = C.c_char_p
        _lib.zk_last_error.restype = C.c_char_p
    return _lib


def check(rc):
    if rc != 0:
        L = lib()
        raise ZkError(rc, "%s -- %s" % (L.zk_strerror(rc).decode(), L.zk_last_error().decode()))
    return rc


def set_device_list(devices):
    """zk_set_device_list: the devices a key uploaded from now on is sharded over, behind ONE handle (include/zkmi355x.h, "multi-device keys").
    An index may repeat (several shards on one card).  Every key handle must have been freed."""
    arr = (C.c_int32 * len(devices))(*[int(d) for d in devices])
    check(lib().zk_set_device_list(arr, C.c_uint32(len(devices))))


def device_list():
    n = C.c_uint32()
    check(lib().zk_get_device_list(None, C.c_uint32(0), C.byref(n)))
    arr = (C.c_int32 * max(1, n.value))()
    check(lib().zk_get_device_list(arr, C.c_uint32(n.value), C.byref(n)))
    return [int(arr[i]) for i in range(n.value)]


def u8(buf):
    """numpy uint8 array / bytes -> (pointer, keepalive)."""
    import numpy as np
    if isinstance(buf, (bytes, bytearray)):
        arr = np.frombuffer(bytes(buf), dtype=np.uint8)
    else:
        arr = np.ascontiguousarray(buf, dtype=np.uint8)
    return arr.ctypes.data_as(C.POINTER(C.c_uint8)), arr
