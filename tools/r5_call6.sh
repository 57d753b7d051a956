timeout -k 5 300 python tools/resident_probe.py --lenses hammer,quincuncial,mercator --size 1920x1080 --frames 400 2>&1 | grep -v amdgpu | cut -c1-330
timeout -k 5 300 python tools/resident_probe.py --lenses hammer --size 2560x1440 --frames 400 2>&1 | grep -v amdgpu | cut -c1-330
timeout -k 2 300 python -m pytest tests/test_apply_resident_gpu.py -x -q -m gpu 2>&1 | tail -2
