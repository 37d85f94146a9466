# Builds the product library (HIP, gfx950) and the test-only oracle.
HIPCC ?= hipcc
ARCH ?= gfx950
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude
CSRC = ronkathon_amd/csrc
LIB = ronkathon_amd/libronk_ntt.so
OBJS = build/tile_kernels.o build/tile_kernels_cfg.o build/ronk_core.o build/ronk_plan.o build/ronk_callers.o build/ronk_dist.o
HDRS = $(wildcard $(CSRC)/*.h) include/ronk_ntt.h

all: $(LIB) oracle
$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)
build/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
oracle:
	$(MAKE) -C oracle -s
clean:
	rm -rf build $(LIB)
	$(MAKE) -C oracle clean
.PHONY: all oracle clean
