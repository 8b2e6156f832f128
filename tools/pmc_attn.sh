set -e
R=$PWD; cd /tmp; export TMPDIR=/tmp
for v in xcd plain; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_attn_$v -o fetch -- $R/tools/probe/attn_ablate_$v 512 > $R/gpurun_out/pmc_attn_$v.log 2>&1
  tail -1 $R/gpurun_out/pmc_attn_$v.log
done
