# Round-3 evidence run, part B: rocprofv3 kernel stats of the bench command, MFMA / HBM PMC passes (separate runs), training-step profile.
set -e
mkdir -p gpurun_out
R=$PWD
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_prof_bench128 -o bench -- python3 $R/bench.py --no-extras --no-cpu-baseline > $R/gpurun_out/r3_bench_B128_under_rocprof.json 2> $R/gpurun_out/r3_rocprof128.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_prof_bench64 -o bench -- python3 $R/bench.py --batch 64 --no-extras --no-cpu-baseline > $R/gpurun_out/r3_bench_B64_under_rocprof.json 2> $R/gpurun_out/r3_rocprof64.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_prof_b1 -o b1 -- python3 $R/tools/prof_nfe.py 1 > $R/gpurun_out/r3_prof_b1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_prof_train_final -o train -- python3 $R/tools/bench_train_full.py 8 > $R/gpurun_out/r3_prof_train_final.log 2>&1
echo "kernel stats done"
cd $R
bash tools/pmc_mfma.sh 64 r03; bash tools/pmc_mfma.sh 128 r03; bash tools/pmc_mfma.sh 1 r03
echo "mfma pmc done"
cd /tmp
for B in 64 128; do for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/r3_pmc_nfe_B${B}_$c -o nfe -- python3 $R/tools/prof_nfe.py $B > $R/gpurun_out/r3_pmc_nfe_B${B}_$c.log 2>&1
done; done
cd $R
for B in 64 128 1; do python tools/pmc_summarize.py gpurun_out/pmc_mfma_r03_B$B > gpurun_out/r3_pmc_mfma_B$B.csv; done
head -8 gpurun_out/r3_pmc_mfma_B64.csv
