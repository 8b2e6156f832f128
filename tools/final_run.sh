set -e
python -m pytest tests -x -q -m gpu > gpurun_out/r2_pytest_gpu.log 2>&1; tail -4 gpurun_out/r2_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err
tail -3 gpurun_out/r2_bench_final.err
cat gpurun_out/r2_bench_final.json
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_prof_final -o bench -- python3 $R/bench.py --no-extras --no-cpu-baseline > $R/gpurun_out/r2_bench_under_rocprof.json 2> $R/gpurun_out/r2_rocprof.err
cat $R/gpurun_out/r2_bench_under_rocprof.json
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_prof_b1 -o b1 -- python3 $R/tools/prof_nfe.py 1 > $R/gpurun_out/r2_prof_b1.log 2>&1
tail -3 $R/gpurun_out/r2_prof_b1.log
