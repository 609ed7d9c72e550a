"""Host-side mirror of the reference's curve plugin `Curve.S` (src/lib/zk/curve.mli:46-54) for the
operations on the prove path, backed by the HIP library.  Names and argument meaning follow the
OCaml modules: `FFT_Fr.fft / ifft / polynomial_mul` (src/lib/zk/FFT.ml:69-105,222-233),
`G.apply_powers`, `G.dot`, `G.powers`, `G.of_Fr` (src/lib/zk/curve.ml:94-118,180).

Values cross this layer in the byte formats of include/zkmi355x.h (Fr 32 B LE; G1 96 B, G2 192 B
uncompressed big-endian).  Error behaviour mirrors the reference: `apply_powers` with fewer
points than coefficients raises ValueError("apply_powers") like `invalid_arg "apply_powers"`
(curve.ml:116); a `dot` over different key sets raises AssertionError (curve.ml:96-100).
"""
import ctypes as C

import numpy as np

from . import _lib

FR_BYTES = 32
ZK_ERR_APPLY_POWERS = -6


def _np(buf):
    if isinstance(buf, np.ndarray):
        return np.ascontiguousarray(buf, dtype=np.uint8).reshape(-1)
    return np.frombuffer(bytes(buf), dtype=np.uint8)


def _p(arr):
    return arr.ctypes.data_as(C.POINTER(C.c_uint8))


class FFT_Fr:
    """FFT.FFT_Fr (src/lib/zk/FFT.ml:222-233): radix-2 NTT over Bls12_381.Fr, natural order."""

    @staticmethod
    def fft(coeffs, log_n=None):
        """fft ?degree f (FFT.ml:69-81): zero-pads to the next power of two, returns N values."""
        a = _np(coeffs)
        n = len(a) // FR_BYTES
        if log_n is None:
            log_n = max(0, (max(n, 1) - 1).bit_length())
        size = 1 << log_n
        if n > size:
            raise ValueError("fft: more coefficients than 2^log_n")
        buf = np.zeros(size * FR_BYTES, dtype=np.uint8)
        buf[:len(a)] = a
        _lib.check(_lib.lib().zk_fr_ntt(_p(buf), C.c_uint32(log_n), 0))
        return buf

    @staticmethod
    def ifft(values):
        """ifft vs (FFT.ml:83-86): inverse transform (the caller normalizes trailing zeros)."""
        a = _np(values).copy()
        n = len(a) // FR_BYTES
        if n == 0 or n & (n - 1):
            raise ValueError("ifft: length must be a power of two")
        _lib.check(_lib.lib().zk_fr_ntt(_p(a), C.c_uint32(n.bit_length() - 1), 1))
        return a

    @staticmethod
    def polynomial_mul(p1, p2):
        """polynomial_mul (FFT.ml:98-105) == Polynomial.mul (polynomial.ml:124-131), normalized."""
        a, b = _np(p1), _np(p2)
        na, nb = len(a) // FR_BYTES, len(b) // FR_BYTES
        out = np.zeros(max(na + nb, 1) * FR_BYTES, dtype=np.uint8)
        nout = C.c_size_t()
        _lib.check(_lib.lib().zk_fr_poly_mul(_p(a), C.c_size_t(na), _p(b), C.c_size_t(nb), _p(out), C.byref(nout)))
        return out[:nout.value * FR_BYTES]


class _Group:
    POINT_BYTES = 0
    COMPRESSED_BYTES = 0
    _msm = _of_fr = _powers = _compress = None

    @classmethod
    def apply_powers(cls, cs, xis, window_bits=0):
        """G.apply_powers cs xis = sum_i cs_i * xis_i (curve.ml:112-118)."""
        c, x = _np(cs), _np(xis)
        out = np.zeros(cls.POINT_BYTES, dtype=np.uint8)
        rc = getattr(_lib.lib(), cls._msm)(_p(x), C.c_size_t(len(x) // cls.POINT_BYTES), _p(c),
                                            C.c_size_t(len(c) // FR_BYTES), C.c_uint32(window_bits), _p(out))
        if rc == ZK_ERR_APPLY_POWERS:
            raise ValueError("apply_powers")
        _lib.check(rc)
        return out

    @classmethod
    def dot(cls, m, c, window_bits=0):
        """G.dot m c = sum_k m_k * c_k over equal key sets (curve.ml:94-103); m, c are dicts
        keyed by variable, or equal-length dense byte vectors."""
        if isinstance(m, dict):
            if set(m) != set(c):
                raise AssertionError("Domain mismatch")
            keys = sorted(m)
            pts = b"".join(bytes(m[k]) for k in keys)
            scs = b"".join(bytes(c[k]) for k in keys)
        else:
            pts, scs = m, c
            if len(_np(pts)) // cls.POINT_BYTES != len(_np(scs)) // FR_BYTES:
                raise AssertionError("Domain mismatch")
        return cls.apply_powers(scs, pts, window_bits)

    @classmethod
    def of_Fr(cls, scalars):
        """G.of_Fr mapped over a vector: one * s_i (curve.ml:180)."""
        s = _np(scalars)
        n = len(s) // FR_BYTES
        out = np.zeros(n * cls.POINT_BYTES, dtype=np.uint8)
        _lib.check(getattr(_lib.lib(), cls._of_fr)(_p(s), C.c_size_t(n), _p(out)))
        return out

    @classmethod
    def powers(cls, d, s):
        """G.powers d s = [g^(s^i) | i = 0..d] (curve.ml:106-109): d+1 points."""
        sc = _np(s)
        out = np.zeros((d + 1) * cls.POINT_BYTES, dtype=np.uint8)
        _lib.check(getattr(_lib.lib(), cls._powers)(C.c_uint32(d), _p(sc), _p(out)))
        return out

    @classmethod
    def to_compressed_bytes(cls, pt):
        p = _np(pt)
        out = np.zeros(cls.COMPRESSED_BYTES, dtype=np.uint8)
        _lib.check(getattr(_lib.lib(), cls._compress)(_p(p), _p(out)))
        return bytes(out)

    _HALF = np.frombuffer(bytes.fromhex("0d0088f51cbff34d258dd3db21a5d66bb23ba5c279c2895fb39869507b587b120f55ffff58a9ffffdcff7fffffffd555"), dtype=np.uint8)   # (p - 1) / 2

    @classmethod
    def to_compressed_bytes_many(cls, pts):
        """to_compressed_bytes (curve.ml:199,208) over a list: pure byte logic (x with the flag bits; the sign bit = the lexicographically leading
        coordinate of y above (p - 1) / 2), vectorised -- what zk_g1/g2_compress do per point.  Uncompressed points back to back in, compressed out."""
        a = np.ascontiguousarray(np.frombuffer(bytes(pts), dtype=np.uint8)).reshape(-1, cls.POINT_BYTES)
        cb = cls.COMPRESSED_BYTES
        out = a[:, :cb].copy()
        inf = (a[:, 0] & 0x40) != 0

        def above_half(y):          # y: (n, 48) big-endian
            diff = y != cls._HALF
            first = diff.argmax(axis=1)
            rows = np.arange(len(y))
            return diff.any(axis=1) & (y[rows, first] > cls._HALF[first])
        if cb == 48:
            large = above_half(a[:, 48:96])
        else:
            y1, y0 = a[:, 96:144], a[:, 144:192]
            large = np.where((y1 != 0).any(axis=1), above_half(y1), above_half(y0))
        out[:, 0] |= 0x80
        out[large, 0] |= 0x20
        out[inf] = 0
        out[inf, 0] = 0xC0
        return out.tobytes()

    @classmethod
    def of_compressed_bytes_many(cls, comp):
        """of_compressed_bytes_exn (curve.ml:199-212) mapped over a list ON THE GPU (zk_g1/g2_decompress_batch): `comp` = the compressed points back
        to back; returns the uncompressed points back to back.  Square roots, curve and subgroup checks per point, as the one-point host call."""
        c = np.ascontiguousarray(np.frombuffer(bytes(comp), dtype=np.uint8))
        n = len(c) // cls.COMPRESSED_BYTES
        if len(c) != n * cls.COMPRESSED_BYTES:
            raise AssertionError("compressed point list: length is not a multiple of the point size")
        out = np.zeros(n * cls.POINT_BYTES, dtype=np.uint8)
        if n:
            _lib.check(getattr(_lib.lib(), cls._decompress_batch)(_p(c), C.c_size_t(n), _p(out)))
        return bytes(out)


class G1(_Group):
    POINT_BYTES, COMPRESSED_BYTES = 96, 48
    _msm, _of_fr, _powers, _compress = "zk_msm_g1", "zk_g1_of_fr", "zk_g1_powers", "zk_g1_compress"
    _decompress_batch = "zk_g1_decompress_batch"


class G2(_Group):
    POINT_BYTES, COMPRESSED_BYTES = 192, 96
    _msm, _of_fr, _powers, _compress = "zk_msm_g2", "zk_g2_of_fr", "zk_g2_powers", "zk_g2_compress"
    _decompress_batch = "zk_g2_decompress_batch"


class GT:
    """Target group element as the 12 Fp coefficients of the tower (576 B, include/zkmi355x.h); the reference's
    GT bytes (curve.ml:217-219) come from its external library, so elements are compared, not serialised alike."""
    BYTES = 576


class Pairing:
    """Curve.S.Pairing (src/lib/zk/curve.mli:46-54) on the host: a handful of pairings per verification."""

    @staticmethod
    def product(g1_points, g2_points):
        """prod_i e(P_i, Q_i) with one final exponentiation -> GT bytes."""
        g1, g2 = bytes(g1_points), bytes(g2_points)
        n = len(g1) // 96
        if len(g1) != 96 * n or len(g2) != 192 * n:
            raise ValueError("pairing product: need as many G1 as G2 points")
        out = C.create_string_buffer(GT.BYTES)
        _lib.check(_lib.lib().zk_pairing_product(g1, g2, C.c_size_t(n), out))
        return out.raw

    @staticmethod
    def pairing(p, q):
        return Pairing.product(bytes(p), bytes(q))

