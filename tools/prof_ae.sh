set -e
R=$PWD; cd /tmp; export TMPDIR=/tmp
for B in 1 8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_prof_ae_B$B -o ae -- python3 $R/tools/prof_ae.py $B > $R/gpurun_out/r2_prof_ae_B$B.log 2>&1
  echo trace B=$B done
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/r2_pmc_ae_$c -o ae -- python3 $R/tools/prof_ae.py 1 4 > $R/gpurun_out/r2_pmc_ae_$c.log 2>&1
  echo pmc $c done
done
