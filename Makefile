# Convenience targets; the driver's entry points are __graft_entry__.build() / smoke(), bench.py and pytest.
PY ?= python

.PHONY: build test-cpu test-gpu bench golden clean

build:            ## hipcc --offload-arch=gfx950 -> tensorflow_end2end_speech_recognition_amd/libasr_hip.so (in-tree)
	$(PY) -m tensorflow_end2end_speech_recognition_amd.build

test-cpu:         ## oracle vs golden / known-answer vectors, ABI, host logic on CPU stand-ins, recipes, gloo data parallel
	$(PY) -m pytest tests -q -m "not gpu"

test-gpu:         ## parity of every HIP op / model through the C ABI (needs an MI355X)
	$(PY) -m pytest tests -q -m gpu

bench:            ## headline: frames/s of 5x256 BLSTM-CTC training on one GPU
	$(PY) bench.py

golden:           ## regenerate tests/golden/* from the reference's own code (needs /root/reference)
	cd /tmp && $(PY) $(CURDIR)/tests/golden/make_golden.py

clean:
	rm -rf tensorflow_end2end_speech_recognition_amd/libasr_hip.so tensorflow_end2end_speech_recognition_amd/csrc/_obj
