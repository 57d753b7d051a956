timeout -k 2 300 python -m pytest tests/test_apply_resident_gpu.py -x -q -m gpu 2>&1 | tail -4 | cut -c1-300
timeout -k 2 120 python tools/r5_dbg3.py 2>&1 | grep -v amdgpu.ids | tail -3
timeout -k 2 900 python -m pytest tests/test_engine_dropin.py tests/test_host_layer.py tests/test_comm_gpu.py -x -q -m gpu 2>&1 | tail -4 | cut -c1-300
