#!/bin/bash
# Kernel experiments: builds pydca_amd/lib/libdca_hip_<name>.so from generator settings given as environment variables,
# then restores the generated sources of the shipped configuration.     tools/build_variant.sh lg80 DCA_GEN_LG21=8,80,1
# Run with DCA_LIB_PATH=pydca_amd/lib/libdca_hip_<name>.so (the SAME library, another inner block; never another backend).
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root"
env "$@" python3 tools/gen_plm_asm.py > /dev/null
make -C pydca_amd/csrc OUT=../lib/libdca_hip_$name.so OBJDIR=../../build/csrc_$name -j8 2>&1 | grep -E "error|warning: v|ran out|spill" || true
python3 tools/gen_plm_asm.py > /dev/null
touch pydca_amd/csrc/logits_gather_asm.inc
ls -la pydca_amd/lib/libdca_hip_$name.so
