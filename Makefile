# Builds the product library (HIP, gfx950) and the test-only oracle.
HIPCC ?= hipcc
ARCH ?= gfx950
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude
CSRC = ronkathon_amd/csrc
LIB = ronkathon_amd/libronk_ntt.so
OBJS = build/tile_kernels_wl.o build/tile_kernels_r4.o build/tile_kernels_mont.o build/tile_kernels_mont_feat.o build/tile_kernels_mont_mul.o build/tile_kernels.o build/tile_kernels_cfg.o build/tile_kernels_half.o build/tile_kernels_feat.o build/tile_kernels_mul.o build/small_kernels.o build/ronk_core.o build/ronk_plan.o build/ronk_callers.o build/ronk_dist.o build/ronk_msm.o
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

# Sanitizer builds of the host-side code: the oracle under ASan+UBSan, the field header and the
# tile kernel body + planner (the host emulator; Goldilocks and Montgomery field policies, the R4 round structure; the scan and long-division bodies) under UBSan (ASan does not follow the emulator's ucontext fibers).
SAN = -g -O1 -fno-omit-frame-pointer -fno-sanitize-recover=all
sanitize:
	@mkdir -p build/san
	gcc $(SAN) -fsanitize=address,undefined -o build/san/oracle_san tests/emu/oracle_san.c oracle/ronk_oracle.c
	g++ $(SAN) -std=c++17 -fsanitize=address,undefined -o build/san/test_gl64_host tests/emu/test_gl64_host.cpp
	gcc -O1 -c -o build/san/orc.o oracle/ronk_oracle.c
	g++ $(SAN) -std=c++17 -fsanitize=undefined -o build/san/emu_tile tests/emu/emu_tile.cpp build/san/orc.o
	g++ $(SAN) -std=c++17 -fsanitize=undefined -o build/san/emu_scan tests/emu/emu_scan.cpp build/san/orc.o
	g++ $(SAN) -std=c++17 -fsanitize=undefined -o build/san/emu_longdiv tests/emu/emu_longdiv.cpp build/san/orc.o
	g++ $(SAN) -std=c++17 -fsanitize=address,undefined -o build/san/bn254_san tests/emu/bn254_san.cpp
	./build/san/oracle_san
	./build/san/bn254_san
	./build/san/test_gl64_host
	./build/san/emu_tile 12 3 0 4 | tail -1
	./build/san/emu_tile 16 2 1 4 18 | tail -1
	./build/san/emu_tile 15 2 0 4 18 25 0 0 1 | tail -1
	./build/san/emu_tile 20 1 0 3 | tail -1
	./build/san/emu_tile dist 16 4 0 0 2 | tail -1
	./build/san/emu_tile mul 20 300001 7 2 18 | tail -1
	RONK_EMU_P=0xFFFFFFFC00000001 RONK_EMU_G=10 ./build/san/emu_tile 16 2 1 4 18 | tail -1
	RONK_EMU_P=0xc0000001 RONK_EMU_G=5 ./build/san/emu_tile 12 3 0 4 | tail -1
	RONK_R4MID=1 ./build/san/emu_tile 19 1 0 4 18 | tail -1
	./build/san/emu_tile 20 1 1 2 20 | tail -1
	RONK_WL_HALF=1 ./build/san/emu_tile 22 1 0 2 22 | tail -1
	RONK_EMU_P=0xFFFFFFFC00000001 RONK_EMU_G=10 ./build/san/emu_tile 20 1 0 2 18 | tail -1
	./build/san/emu_scan 18446744069414584321 70001 123456789 3 1 | tail -1
	./build/san/emu_scan 101 5000 7 3 0 | tail -1
	./build/san/emu_scan 18446744069414584321 70001 123456789 3 3 | tail -1
	EMU_LINDIV1_PL=4 ./build/san/emu_scan 18446744069414584321 30001 123456789 1 6 | tail -1
	./build/san/emu_scan 101 25000 7 3 2 | tail -1
	./build/san/emu_longdiv 18446744069414584321 120 300 64 3 | tail -1
	./build/san/emu_longdiv 101 120 300 64 3 | tail -1
.PHONY: sanitize
