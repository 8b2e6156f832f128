set -e
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_prof_b1 -o b1 -- python3 $R/tools/prof_nfe.py 1 > $R/gpurun_out/r2_prof_b1.log 2>&1
tail -3 $R/gpurun_out/r2_prof_b1.log
python3 $R/tools/sweep_nfe.py 1,2,4,8 
