# Convenience targets; the contract's entry points are __graft_entry__.py (build / smoke), bench.py and pytest.
PY ?= python
.PHONY: build test-cpu test-gpu smoke bench asan-host clean
build:            ## libzkmi355x.so for gfx950 (hipcc cross-compiles without a GPU) and the oracle (the checker; test infrastructure)
	$(PY) -c "import __graft_entry__ as g; g.build()"
test-cpu:         ## oracle against its pins, host logic, ABI, gloo multi-process: no GPU needed
	$(PY) -m pytest tests -x -q -m "not gpu"
test-gpu:         ## parity through the C-ABI on an MI355X (ZK_TEST_FULL=1 adds the 2^20 dense-rows case)
	$(PY) -m pytest tests -x -q -m gpu
smoke:
	$(PY) -c "import __graft_entry__ as g; g.smoke()"
bench:            ## the driver's default line (N = 1); `$(PY) bench.py --gpus N` is its own launcher for N > 1
	$(PY) bench.py
asan-host:        ## the host-only half (pairing, verify, decompress) under AddressSanitizer + UBSan
	$(MAKE) -C zukelang_amd/csrc asan-host
clean:
	$(MAKE) -C zukelang_amd/csrc clean
	$(MAKE) -C oracle clean
