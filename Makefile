# Convenience targets (the driver uses __graft_entry__.build() / pytest / bench.py directly).
PY ?= python

.PHONY: build test-cpu test-gpu smoke bench example clean

build:            ## hipcc --offload-arch=gfx950 -> elliot_amd/csrc/libelliot_hip.so (+ the C oracle); no GPU needed
	$(PY) -c "import __graft_entry__ as g; g.build()"

test-cpu: build   ## oracle vs goldens, ABI, host logic, world-2 gloo
	$(PY) -m pytest tests -q -m "not gpu"

test-gpu: build   ## parity through the C ABI (MI355X)
	$(PY) -m pytest tests -q -m gpu

smoke: build
	$(PY) -c "import __graft_entry__ as g; g.smoke()"

bench: build      ## one JSON line: pairs/s, users/s, rooflines, CPU baseline
	$(PY) bench.py

example: build    ## the C ABI from plain C99
	gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_abi_demo.c -Lelliot_amd/csrc -lelliot_hip \
	    -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$(CURDIR)/elliot_amd/csrc -Wl,-rpath,/opt/rocm/lib -lm -o examples/c_abi_demo
	@echo "run: examples/c_abi_demo   (needs an MI355X)"

clean:
	rm -f elliot_amd/csrc/*.o elliot_amd/csrc/*.so examples/c_abi_demo
	$(MAKE) -C oracle/c clean 2>/dev/null || true
