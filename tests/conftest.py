import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
# before libzkmi355x.so is loaded: the kernel-form switches (ZK_TAIL_SLOTS, ZK_ACC_G1_GLDS, ZK_GRAPH, ...) are cached per process unless this flag is set;
# the GPU suite flips them inside one process to hold every form to the oracle (csrc/zk_common.h: forms_live)
os.environ.setdefault("ZK_TEST_FORMS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")
