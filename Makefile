# Convenience targets (the driver uses __graft_entry__.build(), pytest and bench.py directly).
PY ?= python
.PHONY: build test test-gpu bench smoke clean
build:
	$(PY) -c "import __graft_entry__ as g; g.build()"
test: build
	$(PY) -m pytest tests -q -m "not gpu"
test-gpu: build
	$(PY) -m pytest tests -q -m gpu
smoke: build
	$(PY) -c "import __graft_entry__ as g; g.smoke()"
bench: build
	$(PY) bench.py
clean:
	rm -f mwr_fast_forward_operators_and_lbls_amd/libmwrt.so oracle/liblbl_oracle.so
