#!/bin/bash
# Release build of the library with extra compiler flags, for A/B runs through RALD_LIB_OVERRIDE (tools/ab_libs_nfe.py):
#   tools/build_variant.sh <name> [-DFLAG ...]   ->  rald_amd/librald_hip_<name>.so   (sources: the working tree, or $SRC_REV for a git revision's csrc)
set -e
name=$1; shift
rm -rf /tmp/variant_$name && mkdir -p /tmp/variant_$name/rald_amd && cp -r /root/repo/include /tmp/variant_$name/
if [ -n "$SRC_REV" ]; then (cd /root/repo && git archive $SRC_REV rald_amd/csrc | tar -x -C /tmp/variant_$name); else cp -r /root/repo/rald_amd/csrc /tmp/variant_$name/rald_amd/; fi
rm -rf /tmp/variant_$name/rald_amd/csrc/build /tmp/variant_$name/rald_amd/csrc/build_probe
cd /tmp/variant_$name/rald_amd/csrc
make -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form=1 $*" 2>&1 | grep -E "error|Error" || true
cp /tmp/variant_$name/rald_amd/librald_hip.so /root/repo/rald_amd/librald_hip_$name.so
ls -la /root/repo/rald_amd/librald_hip_$name.so
