set -e
R=$PWD; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_enc_$c -o enc -- python3 $R/tools/prof_enc.py 4 > $R/gpurun_out/pmc_enc_$c.log 2>&1
  echo done $c
done
