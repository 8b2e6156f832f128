set -e
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b8 -o b8 -- python3 $R/tools/prof_nfe.py 8 > $R/gpurun_out/prof_b8.log 2>&1
cat $R/gpurun_out/prof_b8.log | tail -3
