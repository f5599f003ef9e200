#!/bin/bash
# ISA of one csrc/*.hip for gfx950: tools/isa.sh fftconv_kernels > /tmp/fc.s
set -e
src="$(dirname "$0")/../diffwave-sashimi_amd/csrc/$1.hip"
extra=""; [ "$1" = fftconv_kernels ] && extra="-fno-slp-vectorize"   # build.py FILE_FLAGS
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $extra --cuda-device-only -S -o - "$src"
