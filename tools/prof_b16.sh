set -e
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b16 -o b16 -- python3 $R/tools/prof_nfe.py 16 > $R/gpurun_out/prof_b16.log 2>&1
