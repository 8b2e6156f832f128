set -e
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_enc -o enc -- python3 $R/tools/prof_enc.py 8 > $R/gpurun_out/prof_enc.log 2>&1
echo done
