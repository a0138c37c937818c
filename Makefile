# Native build without Python: the C-ABI library (HIP kernels + C shim), the raw-frame CLI and
# the CPU oracle.  `python -m ffmpeg_ffv2_amd.build` does the same for the library.
HIPCC   ?= hipcc
CC      ?= gcc
HIPFLAGS = --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-strict-aliasing -ffp-contract=off -Wall
CSRC     = ffmpeg_ffv2_amd/csrc
LIB      = ffmpeg_ffv2_amd/libffv2amd.so
OBJS     = $(CSRC)/ffv2_kernels.o $(CSRC)/ffv2_pvq.o $(CSRC)/ffv2_inverse.o $(CSRC)/ffv2_upconv.o $(CSRC)/ffv2_wide.o $(CSRC)/ffv2_rangecoder.o $(CSRC)/ffv2_lanecoder.o $(CSRC)/ffv2_capi.o $(CSRC)/ffv2enc_amd.o $(CSRC)/ffv2mkv.o

all: $(LIB) examples/ffv2enc_cli oracle

gen:
	python3 tools/gen_code.py

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/ffv2_kernels.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/ffv2_capi.o: $(CSRC)/ffv2_capi.cpp $(CSRC)/ffv2_kernels.h include/ffv2_amd.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/ffv2mkv.o: $(CSRC)/ffv2mkv.c include/ffv2_amd_mkv.h
	$(CC) -O2 -fPIC -std=gnu11 -Wall -Iinclude -c $< -o $@

$(CSRC)/ffv2enc_amd.o: $(CSRC)/ffv2enc_amd.c include/ffv2_amd.h include/ffv2_amd_codec.h
	$(CC) -O2 -fPIC -std=gnu11 -Wall -Iinclude -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(OBJS)

examples/ffv2enc_cli: examples/ffv2enc_cli.c $(LIB)
	$(CC) -O2 -std=gnu11 -Wall -Iinclude $< -o $@ -Lffmpeg_ffv2_amd -lffv2amd -Wl,-rpath,'$$ORIGIN/../ffmpeg_ffv2_amd'

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(OBJS) $(LIB) examples/ffv2enc_cli
	$(MAKE) -C oracle clean

.PHONY: all gen oracle clean
