set -e
R=$PWD; cd /tmp; export TMPDIR=/tmp
for v in 128 256 384; do
export RALD_GEMM_ABLATE=$v
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_nfe_pol$v -o nfe -- python3 $R/tools/prof_nfe.py 64 > $R/gpurun_out/pmc_nfe_pol$v.log 2>&1
echo done $v
done
