set -e
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_prof_b64 -o b64 -- python3 $R/tools/prof_nfe.py 64 > $R/gpurun_out/r2_prof_b64.log 2>&1
tail -2 $R/gpurun_out/r2_prof_b64.log
