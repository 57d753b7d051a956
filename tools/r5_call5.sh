timeout -k 5 100 python tools/r5_hostcost.py 2>&1 | grep HOSTCOST
timeout -k 5 300 python tools/r5_stripes.py 2>&1 | grep STRIPES
timeout -k 5 100 python tools/r5_lat.py panini 0 2>&1 | grep LAT
timeout -k 2 300 python -m pytest tests/test_apply_resident_gpu.py -x -q -m gpu 2>&1 | tail -3 | cut -c1-300
for F in 0; do timeout -k 2 90 python tools/r5_stress.py $F 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-150; done
