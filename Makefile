# Builds the gfx950 engine (libslacken_amd.so) in-tree, and the CPU oracle used by the tests.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
WPS ?= 0
LWPS ?= 0
LLW ?= 4
EXTRA ?=
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function -DSLK_WPS=$(WPS) -DSLK_LANE_WPS=$(LWPS) -DSLK_LANE_LW=$(LLW) $(EXTRA)
CSRC := slacken_amd/csrc
LIB := slacken_amd/lib/libslacken_amd.so

CLI := slacken_amd/bin/slacken-amd

GATHER := slacken_amd/lib/libslk_gather.so

all: $(LIB) $(CLI) $(GATHER) oracle

# measurement helper of bench.py (the part's random-request rate, measured in the bench run): not linked or loaded by the product
$(GATHER): tools/gather_rate.hip
	@mkdir -p slacken_amd/lib
	$(HIPCC) -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -shared -o $@ tools/gather_rate.hip

$(LIB): $(CSRC)/kernels.hip $(CSRC)/fused.hip $(CSRC)/lane.hip $(CSRC)/build.hip $(CSRC)/shard.hip $(CSRC)/wide.hip $(CSRC)/capi.hip $(CSRC)/shardset.hip $(CSRC)/engine.h $(CSRC)/hostside.h slacken_amd/host/pack.hpp include/slacken_amd.h
	@mkdir -p slacken_amd/lib
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/kernels.hip $(CSRC)/fused.hip $(CSRC)/lane.hip $(CSRC)/build.hip $(CSRC)/shard.hip $(CSRC)/wide.hip $(CSRC)/capi.hip $(CSRC)/shardset.hip -ldl

# Parquet input of the CLI: the Arrow C++ libraries inside the pyarrow wheel, if there is one (no Arrow dev package here)
PYARROW_DIR := $(shell python3 -c "import pyarrow, os; print(os.path.dirname(pyarrow.__file__))" 2>/dev/null)
ifneq ($(wildcard $(PYARROW_DIR)/libparquet.so.*),)
PQ_LIB := $(notdir $(firstword $(sort $(wildcard $(PYARROW_DIR)/libparquet.so.[0-9]*))))
AR_LIB := $(notdir $(firstword $(sort $(wildcard $(PYARROW_DIR)/libarrow.so.[0-9]*))))
PQ_CXXFLAGS := -DSLK_HAVE_PARQUET -I$(PYARROW_DIR)/include
PQ_LDFLAGS := -L$(PYARROW_DIR) -l:$(PQ_LIB) -l:$(AR_LIB) -Wl,-rpath,$(PYARROW_DIR)
endif

slacken_amd/bin/parquet_source.o: slacken_amd/host/parquet_source.cpp slacken_amd/host/parquet_source.hpp
	@mkdir -p slacken_amd/bin
	g++ -O2 -std=c++20 -Wall $(PQ_CXXFLAGS) -c -o $@ slacken_amd/host/parquet_source.cpp

$(CLI): slacken_amd/bin/parquet_source.o slacken_amd/host/slacken_cli.cpp slacken_amd/host/taxonomy.hpp slacken_amd/host/seqio.hpp slacken_amd/host/pargz.hpp slacken_amd/host/parbz2.hpp slacken_amd/host/titles.hpp slacken_amd/host/output.hpp slacken_amd/host/pack.hpp include/slacken_amd.h $(LIB)
	@mkdir -p slacken_amd/bin
	g++ -O2 -std=c++17 -Wall -o $@ slacken_amd/host/slacken_cli.cpp slacken_amd/bin/parquet_source.o $(PQ_LDFLAGS) -Lslacken_amd/lib -lslacken_amd -lz -ldl -lpthread -Wl,-rpath,'$$ORIGIN/../lib'

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(LIB) $(GATHER)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
