# Convenience targets (the driver uses __graft_entry__.build() / pytest / bench.py directly).
all: build
build:
	$(MAKE) -C 3dscanner-graycode_amd -j8
	$(MAKE) -C oracle
test:            ## CPU suite (oracle vs goldens, host logic, ABI) incl. the sanitizer runs of the native CPU code
	$(MAKE) -C oracle san
	python -m pytest tests -x -q -m "not gpu"
san:             ## ASan + UBSan + TSan on the CPU builds (host_util.cpp unit tests, C oracle through the golden vectors)
	$(MAKE) -C oracle san-run
test-gpu:        ## needs an MI355X
	python -m pytest tests -x -q -m gpu
bench:
	python bench.py
golden:          ## regenerate tests/golden/*.npz by running the reference (build container only)
	python tests/golden/make_golden.py
clean:
	$(MAKE) -C 3dscanner-graycode_amd clean
	$(MAKE) -C oracle clean
.PHONY: all build test san test-gpu bench golden clean
