timeout -k 5 200 python tools/r5_rubix.py 2>&1 | grep RUBIX
timeout -k 5 400 python -m pytest tests/test_apply_gpu.py tests/test_apply_resident_gpu.py tests/test_saveglobe.py -x -q -m gpu 2>&1 | tail -2
