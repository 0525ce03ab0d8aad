# Convenience targets; __graft_entry__.build() does the same three builds.
all: lib oracle host
lib:
	$(MAKE) -C raytracing_amd/csrc
oracle:
	$(MAKE) -C oracle
host: lib
	$(MAKE) -C raytracing_amd/host
diagnostics: lib
	$(MAKE) -C raytracing_amd/csrc prof
test:
	python -m pytest tests -q -m "not gpu"
test-gpu:
	python -u -m pytest tests -q -m gpu --timeout 120 --timeout-method thread
.PHONY: all lib oracle host diagnostics test test-gpu
