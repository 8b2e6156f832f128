set -e
R=$PWD; cd /tmp; export TMPDIR=/tmp
for B in 1 8; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_prof_enc$B -o enc -- python3 $R/tools/prof_enc_ae.py $B > $R/gpurun_out/r2_prof_enc$B.log 2>&1
done
