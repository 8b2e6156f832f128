set -e
R=$PWD; cd /tmp; export TMPDIR=/tmp
for v in old new; do
  if [ $v = old ]; then export RALD_LIB_OVERRIDE=$R/rald_amd/librald_hip_old.so; else unset RALD_LIB_OVERRIDE; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_prof_$v -o b64 -- python3 $R/tools/prof_nfe.py 64 > $R/gpurun_out/r3_prof_$v.log 2>&1
  head -12 $R/gpurun_out/r3_prof_$v/b64_kernel_stats.csv | cut -c1-200
done
