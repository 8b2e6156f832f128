#!/bin/bash
# usage: tools/asm.sh <file.hip under rald_amd/csrc> <mangled-name substring>  -> /tmp/k.s (that kernel), register summary, spill sites
cd /root/repo/rald_amd/csrc || exit 1
f=$1; k=$2
hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -S --cuda-device-only $ASMFLAGS $f -o /tmp/all.s 2>&1 | grep -E "error" | head
grep -E "\.name:|\.vgpr_count|\.vgpr_spill_count|\.sgpr_count" /tmp/all.s | grep -A3 "$k" | head -8
name=$(grep -E "^_Z.*$k.*:" /tmp/all.s | head -1 | sed 's/:.*//')
awk -v n="$name:" '$1==n{f=1} f{print} f&&/s_endpgm/{exit}' /tmp/all.s > /tmp/k.s
echo "kernel $name: $(wc -l < /tmp/k.s) lines; scratch ops: $(grep -c scratch_ /tmp/k.s)"
grep -n "scratch_\|s_barrier\|Loop Header" /tmp/k.s | awk '{print $1, $2}' | head -${3:-30}
