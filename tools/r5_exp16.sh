set -x
mkdir -p gpurun_out/r5p
python -m pytest tests/test_general_gpu.py -x -q > gpurun_out/r5p/test_general.txt 2>&1
python -m pytest tests/test_wino_gpu.py tests/test_parity_gpu.py -x -q -k "not bench" > gpurun_out/r5p/test_rest.txt 2>&1
