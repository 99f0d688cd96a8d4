# Builds everything in-tree for gfx950 (MI355X):
#   polypolish_amd/_build/libpolypolish_hip.so   the product: HIP kernels + C ABI + host ingest
#   bin/polypolish                               the drop-in CLI (links the library)
#   oracle/_build/*                              the CPU oracle (test infrastructure only)
#   bin/polish_min                               examples/polish_min.c: a plain C99 host over the C ABI
#   tools/_build/libsamgen.so                    SAM text writer for bench.py's end-to-end leg (not product)
# -ffp-contract=off: the vote's banker's rounding must see the unfused product depth*fraction.
HIPCC    ?= hipcc
ARCH     ?= gfx950
CSRC     := polypolish_amd/csrc
OUT      := polypolish_amd/_build
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fPIC -Iinclude -I$(CSRC) -Wall -Wno-unused-function

LIB  := $(OUT)/libpolypolish_hip.so
OBJS := $(OUT)/pp_kernels.o $(OUT)/pp_filter.o $(OUT)/pp_tokenize.o $(OUT)/pp_filter_dev.o $(OUT)/pp_comm.o $(OUT)/pp_shard_dev.o $(OUT)/pp_ingest.o $(OUT)/pp_driver.o $(OUT)/pp_filter_host.o $(OUT)/pp_shard.o

all: $(LIB) bin/polypolish bin/polish_min oracle tools/_build/libsamgen.so

$(OUT)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/polypolish_hip.h
	@mkdir -p $(OUT)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OUT)/%.o: $(CSRC)/%.cpp include/polypolish_hip.h $(CSRC)/pp_host.h
	@mkdir -p $(OUT)
	$(HIPCC) $(HIPFLAGS) -x c++ -c $< -o $@

# pp_kernels.hip without MachineLICM: the pass hoists constants and thread-number arithmetic out of k_tile's item loops and
# the allocator then spills what the loops need (k_tile with a whole read per lane: 14 spilled VGPRs with it, reloaded
# inside the pass; without it the few spills left sit outside the loops).  Same speed for every other kernel (measured).
KFLAGS ?= -mllvm -disable-machine-licm
$(OUT)/pp_kernels.o: HIPFLAGS += $(KFLAGS)

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -lz -lpthread -ldl

bin/polypolish: $(CSRC)/pp_cli.cpp $(CSRC)/pp_host.h $(LIB)
	@mkdir -p bin
	$(HIPCC) -O2 -std=c++17 -Iinclude -I$(CSRC) -x c++ $(CSRC)/pp_cli.cpp -o $@ -L$(OUT) -lpolypolish_hip -Wl,-rpath,'$$ORIGIN/../$(OUT)'

# strict C99: the header is a C header, and nothing but the C ABI is needed on the caller's side
bin/polish_min: examples/polish_min.c include/polypolish_hip.h $(LIB)
	@mkdir -p bin
	gcc -std=c99 -O2 -Wall -Wextra -pedantic -Werror -Iinclude $< -o $@ -L$(OUT) -lpolypolish_hip -Wl,-rpath,'$$ORIGIN/../$(OUT)'

oracle:
	$(MAKE) -C oracle

# SAM / FASTA text writer of the synthetic workloads (bench.py's end-to-end leg; measurement infrastructure)
tools/_build/libsamgen.so: tools/samgen.c
	@mkdir -p tools/_build
	gcc -O2 -std=c11 -Wall -Wextra -shared -fPIC $< -o $@

# kernel experiments: make variant NAME=x DEFS="-DPP_..." -> $(OUT)/var_x/libpolypolish_hip.so (use with PP_LIB_PATH)
variant: $(LIB)
	@mkdir -p $(OUT)/var_$(NAME)
	$(HIPCC) $(HIPFLAGS) $(KFLAGS) $(DEFS) -c $(CSRC)/pp_kernels.hip -o $(OUT)/var_$(NAME)/pp_kernels.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $(OUT)/var_$(NAME)/libpolypolish_hip.so $(OUT)/var_$(NAME)/pp_kernels.o $(filter-out $(OUT)/pp_kernels.o,$(OBJS)) -lz -lpthread -ldl

# the host side (parsers, filter loader and writer, planner, drivers) under AddressSanitizer; the device objects as they are:
#   make asan && LD_PRELOAD=$$(hipcc -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 \
#       PP_LIB_PATH=$(OUT)/asan/libpolypolish_hip.so python -m pytest tests -m "not gpu" -q
HOSTSRC := pp_ingest pp_filter_host pp_shard pp_driver
asan: $(LIB)
	@mkdir -p $(OUT)/asan
	for f in $(HOSTSRC); do $(HIPCC) --offload-arch=$(ARCH) -O1 -g -std=c++17 -ffp-contract=off -fPIC -Iinclude -I$(CSRC) \
	    -fsanitize=address -fno-omit-frame-pointer -x c++ -c $(CSRC)/$$f.cpp -o $(OUT)/asan/$$f.o || exit 1; done
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -fsanitize=address -o $(OUT)/asan/libpolypolish_hip.so \
	    $(filter-out $(addprefix $(OUT)/,$(addsuffix .o,$(HOSTSRC))),$(OBJS)) $(addprefix $(OUT)/asan/,$(addsuffix .o,$(HOSTSRC))) -lz -lpthread -ldl

clean:
	rm -rf $(OUT) bin oracle/_build tools/_build

.PHONY: all oracle clean variant asan
