timeout -k 2 150 python -m pytest tests/test_apply_resident_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout -k 2 100 python tools/r5_diag.py --lenses stereographic,hammer,panini --size 1920x1080 --flags 0,16 2>&1 | grep DIAG
timeout -k 2 100 python tools/r5_diag.py --lenses panini,hammer --flags 0 2>&1 | grep DIAG
for F in 0; do timeout -k 2 90 python tools/r5_stress.py $F 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-200; done
