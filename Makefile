# Convenience targets (the driver uses __graft_entry__.build() / pytest / bench.py directly).
.PHONY: build test test-gpu bench clean
build:
	python -c "import __graft_entry__ as g; g.build()"
test: build
	python -m pytest tests -q -m "not gpu"
test-gpu: build
	python -m pytest tests -q -m gpu
bench: build
	python bench.py
clean:
	rm -rf rectipy_amd/_C oracle/_build gpurun_out
