#!/usr/bin/env python3
"""Round 5: rate of zk_g1/g2_decompress_batch (of_compressed_bytes_exn over a list, on the GPU) against the one-point host calls."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from zukelang_amd import _lib, r1cs as RC
from zukelang_amd.curve import G1, G2
_lib.check(_lib.lib().zk_init(0))
for grp, logn in ((G1, 20), (G2, 18)):
    n = 1 << logn
    pts = grp.of_Fr(RC.random_fr_bytes(n, 7))
    B, Cb = grp.POINT_BYTES, grp.COMPRESSED_BYTES
    k = 2000
    comp_small = b"".join(grp.to_compressed_bytes(pts[B * i:B * (i + 1)]) for i in range(k))
    # tile the 2000 compressed points up to n (compression of 2^20 points one by one on the host is what this script is about NOT doing)
    comp = (comp_small * (n // k + 1))[:Cb * n]
    grp.of_compressed_bytes_many(comp[:Cb * 4096])
    t = time.perf_counter(); out = grp.of_compressed_bytes_many(comp); dt = time.perf_counter() - t
    assert out[:B * k] == bytes(pts[:B * k])
    from zukelang_amd import wire
    t = time.perf_counter()
    for i in range(200):
        (wire.g1_of_json if grp is G1 else wire.g2_of_json)(comp[Cb * i:Cb * (i + 1)])
    host = (time.perf_counter() - t) / 200
    print("%s: 2^%d points in %.3f s on the GPU (%.2f M points/s, H2D + D2H included); host one-point call %.3f ms -> %.0f s for the same list" % (
        grp.__name__, logn, dt, n / dt / 1e6, host * 1e3, host * n), flush=True)
