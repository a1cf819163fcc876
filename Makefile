# Builds what __graft_entry__.build() builds, for maintainers who do not go through Python:
#   make            the gfx950 library, the C++ host layer's cluster example, the stream generator, the CPU oracle (checker)
#   make test-cpu   the CPU test suite (oracle KATs, ABI, host logic, gloo multi-rank)
# hipcc cross-compiles for gfx950 without a GPU.
HIPCC ?= hipcc
CXX ?= g++
CSRC := gigapaxos_amd/csrc
HOST := gigapaxos_amd/host
LIB := $(CSRC)/libgpx_hip.so
CLUSTER := $(HOST)/gpx_loopback_cluster
STREAMS := gigapaxos_amd/native/libgpx_streams.so

all: $(LIB) $(CLUSTER) $(STREAMS) oracle

# SURVEY.md 8(d)'s synthetic stream generator (host only, plain C): what bench.py fills its rounds with
$(STREAMS): gigapaxos_amd/native/gpx_streams.c
	$(CC) -O2 -shared -fPIC -Wall -Wextra -o $@ $<

$(LIB): $(wildcard $(CSRC)/*.hip $(CSRC)/*.h $(CSRC)/*.inc) include/gpx.h include/gpx_wire.h
	cd $(CSRC) && $(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result -o libgpx_hip.so gpx_engine.hip

$(CLUSTER): $(HOST)/gpx_host.cpp $(HOST)/loopback_cluster.cpp $(HOST)/gpx_host.hpp $(LIB)
	$(CXX) -O2 -std=c++17 -pthread -Wall -Wextra -Iinclude -o $@ $(HOST)/gpx_host.cpp $(HOST)/loopback_cluster.cpp \
	    -L$(CSRC) -lgpx_hip '-Wl,-rpath,$$ORIGIN/../csrc' -Wl,-rpath-link,/opt/rocm/lib

oracle:
	$(MAKE) -C oracle -s

test-cpu: all
	python -m pytest tests -q -m "not gpu"

clean:
	rm -f $(LIB) $(CLUSTER) $(STREAMS) oracle/libgpx_oracle.so oracle/_host_cluster_oracle

.PHONY: all oracle test-cpu clean
