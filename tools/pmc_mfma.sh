# MFMA-utilisation counters over whole NFEs (every kernel in situ): rocprofv3 --pmc in its own run with --kernel-trace only.
# usage: bash tools/pmc_mfma.sh <batch> <tag>
set -e
R=$PWD; B=${1:-64}; TAG=${2:-r03}
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv \
  -d $R/gpurun_out/pmc_mfma_${TAG}_B$B -o nfe -- python3 $R/tools/prof_nfe.py $B > $R/gpurun_out/pmc_mfma_${TAG}_B$B.log 2>&1
echo "pmc mfma B=$B done"
