set -e
python -m pytest tests/test_gpu_boundary.py -q -m gpu 2>&1 | tail -3
python tools/ab_two_stream.py 128,256 2>&1 | tee gpurun_out/r3_two_stream.txt
