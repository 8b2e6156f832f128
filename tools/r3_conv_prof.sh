set -e
R=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_prof_enc -o enc -- python3 $R/tools/prof_enc.py 8 > $R/gpurun_out/r3_prof_enc.log 2>&1
head -12 $R/gpurun_out/r3_prof_enc/enc_kernel_stats.csv | cut -c1-150
