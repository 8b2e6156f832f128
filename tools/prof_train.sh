set -e
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_prof_train -o train -- python3 $R/tools/bench_train_full.py 8 > $R/gpurun_out/r2_prof_train.log 2>&1
tail -3 $R/gpurun_out/r2_prof_train.log
