#!/bin/bash
# Build a named variant of the library with extra compiler flags into amaranth_twstft_amd/variants/
# (git-ignored; travels with gpurun):   tools/variants.sh noload "-DTWX_ABL=4"
# Run on the GPU box:                   TWX_LIB=amaranth_twstft_amd/variants/lib_noload.so python bench.py ...
set -e
name=$1; shift
cd "$(dirname "$0")/../amaranth_twstft_amd/csrc"
mkdir -p ../variants
make -j8 OBJDIR=build_$name OUT=../variants/lib_$name.so EXTRA="$*" 2>&1 | grep -E "error|Error" || true
ls -la ../variants/lib_$name.so
