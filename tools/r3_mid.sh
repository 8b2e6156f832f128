# Round-3: kernel stats at mid batches (B = 8, 16, 32) and of the training step.
set -e
R=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
for B in 8 16 32; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_prof_b$B -o b$B -- python3 $R/tools/prof_nfe.py $B > $R/gpurun_out/r3_prof_b$B.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_prof_train -o train -- python3 $R/tools/bench_train_full.py 8 > $R/gpurun_out/r3_prof_train.log 2>&1
tail -3 $R/gpurun_out/r3_prof_train.log
cd $R; python tools/sweep_nfe.py > gpurun_out/r3_sweep_nfe.log 2>&1; tail -12 gpurun_out/r3_sweep_nfe.log
