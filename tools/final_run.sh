set -e
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -3 gpurun_out/bench_final.err
cat gpurun_out/bench_final.json
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final5 -o bench -- python3 $R/bench.py --no-extras --no-cpu-baseline > $R/gpurun_out/bench_under_rocprof.json 2> $R/gpurun_out/rocprof.err
cat $R/gpurun_out/bench_under_rocprof.json
