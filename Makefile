# convenience targets; the contract entry points are __graft_entry__.py (build/smoke), tests/ and bench.py
PY ?= python

build:            ## hipcc gfx950 library + host tools + oracle (+ genuine reference when /root/reference exists)
	$(PY) -c "import __graft_entry__ as g; g.build()"

test:             ## CPU suite (oracle pinning, loader/writer/ABI, gloo sharding)
	$(PY) -m pytest tests -x -q -m "not gpu"

test-gpu:         ## bit-exact parity on the MI355X
	$(PY) -m pytest tests -x -q -m gpu

bench:
	$(PY) bench.py

soak:             ## randomised parity soak, 40 indexes of 0.3-6 M rows (GPU box)
	$(PY) tools/soak.py 40

.PHONY: build test test-gpu bench soak
