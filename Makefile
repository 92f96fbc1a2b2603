# Build recipe (no cmake needed): product library (HIP, gfx950), synthetic-frame helper (C),
# and the CPU oracle (C++, test infrastructure).  `python -c "import __graft_entry__ as g; g.build()"` runs this.
HIPCC      ?= /opt/rocm/bin/hipcc
CXX        ?= g++
CC         ?= gcc
ARCH       ?= gfx950
# -amdgpu-mfma-vgpr-form: the matcher's MFMA results feed VALU min / compare trees, so they should land in VGPRs (no v_accvgpr_read per value)
HIPFLAGS   := --offload-arch=$(ARCH) -O3 -std=c++20 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -Wall -Wno-unused-function -Iinclude -Iorb_slam_amd/csrc
ORBX_SRCS  := $(wildcard orb_slam_amd/csrc/*.hip)
ORBX_HDRS  := $(wildcard orb_slam_amd/csrc/*.h orb_slam_amd/csrc/*.inc include/*.h)

all: orb_slam_amd/liborbx.so orb_slam_amd/libsynthframes.so oracle/liborb_oracle.so oracle_ref orb_slam_amd/cpp/example_frame orb_slam_amd/cpp/example_pipeline orb_slam_amd/cpp/example_lanes orb_slam_amd/cpp/bench_single_frame tools/microbench/valu_rate tools/microbench/valu_rate2 tools/microbench/mfma_layout tools/microbench/mfma_layout_fp4 tools/microbench/mfma_valu_mix tools/microbench/fetch_calib tools/microbench/valu_exec_mask tools/microbench/ta_shapes

# the hash of the kernel sources travels inside the library (orbx_build_id): counters replayed by bench.py must come from THIS build
SRC_HASH   := $(shell cat $(sort $(ORBX_SRCS) $(ORBX_HDRS)) | sha256sum | cut -c1-16)
# one object per translation unit (round 6: the extractor's kernels are split by stage; `make -j` compiles them side by side).  The hash goes into
# the one file that reports it, which therefore depends on every source.
OBJDIR     := build/orbx
ORBX_OBJS  := $(patsubst orb_slam_amd/csrc/%.hip,$(OBJDIR)/%.o,$(ORBX_SRCS))
$(OBJDIR)/%.o: orb_slam_amd/csrc/%.hip $(ORBX_HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(OBJDIR)/orbx_api.o: orb_slam_amd/csrc/orbx_api.hip $(ORBX_SRCS) $(ORBX_HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -DORBX_SRC_HASH='"$(SRC_HASH)"' -c $< -o $@
orb_slam_amd/liborbx.so: $(ORBX_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -fPIC -shared $(ORBX_OBJS) -o $@

orb_slam_amd/libsynthframes.so: orb_slam_amd/csrc/synth_frames.c
	$(CC) -O2 -fPIC -shared $< -o $@

# oracle: scalar restatement, ISO float evaluation (no FMA contraction), the CPU baseline build flags of SURVEY §8d
oracle/liborb_oracle.so: oracle/orb_oracle.cpp oracle/bow_oracle.cpp oracle/frame_oracle.cpp oracle/search_oracle.cpp oracle/orb_pattern_points.inc
	$(MAKE) -C oracle liborb_oracle.so

# the reference's own sources compiled against stand-in cv headers (only where /root/reference exists): oracle/Makefile
oracle_ref: oracle/liborb_oracle.so orb_slam_amd/liborbx.so
	$(MAKE) -C oracle ref

# C++ shim demo: the reference's Frame-side call sequence against the drop-in classes (host C++, links the C ABI)
orb_slam_amd/cpp/example_frame: orb_slam_amd/cpp/example_frame.cpp orb_slam_amd/cpp/ORBextractor.h orb_slam_amd/cpp/ORBmatcher.h orb_slam_amd/cpp/ORBVocabulary.h orb_slam_amd/cpp/cvcompat.h include/orbx.h include/orbv.h orb_slam_amd/liborbx.so
	$(CXX) -O2 -std=c++14 -Iinclude -Iorb_slam_amd/cpp $< -o $@ -Lorb_slam_amd -lorbx -Wl,-rpath,'$$ORIGIN/..' -Wl,-rpath,/opt/rocm/lib

# the device-resident front-end driven from plain C++ through the C ABI (no HIP headers on the host side)
orb_slam_amd/cpp/example_pipeline: orb_slam_amd/cpp/example_pipeline.cpp include/orbx.h include/orbf.h include/orbv.h include/orbs.h orb_slam_amd/liborbx.so
	$(CXX) -O2 -std=c++14 -Iinclude $< -o $@ -Lorb_slam_amd -lorbx -Wl,-rpath,'$$ORIGIN/..' -Wl,-rpath,/opt/rocm/lib

# the lanes configuration (LanePipeline.h) from plain C++
orb_slam_amd/cpp/example_lanes: orb_slam_amd/cpp/example_lanes.cpp orb_slam_amd/cpp/LanePipeline.h include/orbx.h orb_slam_amd/liborbx.so
	$(CXX) -O2 -std=c++14 -Iinclude -Iorb_slam_amd/cpp $< -o $@ -Lorb_slam_amd -lorbx -Wl,-rpath,'$$ORIGIN/..' -Wl,-rpath,/opt/rocm/lib

# the drop-in call's latency from plain C++
orb_slam_amd/cpp/bench_single_frame: orb_slam_amd/cpp/bench_single_frame.cpp include/orbx.h orb_slam_amd/liborbx.so
	$(CXX) -O2 -std=c++14 -Iinclude $< -o $@ -Lorb_slam_amd -lorbx -Wl,-rpath,'$$ORIGIN/..' -Wl,-rpath,/opt/rocm/lib

# measurement aid: issue rate of the VALU opcodes the kernels are made of (profiles/r01_valu_issue_rates.txt)
tools/microbench/valu_rate: tools/microbench/valu_rate.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -Wno-unused-value $< -o $@
tools/microbench/valu_rate2: tools/microbench/valu_rate2.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -Wno-unused-value $< -o $@
tools/microbench/mfma_layout: tools/microbench/mfma_layout.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -Wno-unused-result $< -o $@
tools/microbench/mfma_layout_fp4: tools/microbench/mfma_layout_fp4.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -Wno-unused-value $< -o $@
tools/microbench/mfma_valu_mix: tools/microbench/mfma_valu_mix.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -Wno-unused-result $< -o $@
tools/microbench/fetch_calib: tools/microbench/fetch_calib.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -Wno-unused-result $< -o $@
# round 4: VALU issue cost against the EXEC mask; vector-memory cost against the access shape (profiles/r04_valu_exec_mask.txt, r04_ta_shapes.txt)
tools/microbench/valu_exec_mask: tools/microbench/valu_exec_mask.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -Wno-unused-value $< -o $@
tools/microbench/ta_shapes: tools/microbench/ta_shapes.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++20 -Wno-unused-value $< -o $@

clean:
	rm -f tools/microbench/mfma_layout_fp4 tools/microbench/valu_exec_mask tools/microbench/ta_shapes tools/microbench/valu_rate tools/microbench/valu_rate2 tools/microbench/mfma_layout tools/microbench/mfma_valu_mix tools/microbench/fetch_calib orb_slam_amd/cpp/bench_single_frame orb_slam_amd/cpp/example_lanes orb_slam_amd/cpp/example_frame orb_slam_amd/cpp/example_pipeline orb_slam_amd/liborbx.so orb_slam_amd/libsynthframes.so oracle/liborb_oracle.so
	rm -rf oracle/_ref oracle/_ref_native build/orbx

.PHONY: all clean oracle_ref
