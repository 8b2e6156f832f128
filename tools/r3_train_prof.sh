set -e
R=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_prof_train2 -o train -- python3 $R/tools/bench_train_full.py 8 > $R/gpurun_out/r3_prof_train2.log 2>&1
grep "ms/step" $R/gpurun_out/r3_prof_train2.log
