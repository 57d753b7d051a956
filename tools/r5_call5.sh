timeout -k 2 300 python -m pytest tests/test_apply_resident_gpu.py tests/test_apply_gpu.py -x -q -m gpu 2>&1 | tail -3 | cut -c1-300
