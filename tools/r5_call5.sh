for F in 0 0 0 0; do timeout -k 2 90 python tools/r5_stress.py $F 2>&1 | grep -v amdgpu.ids | tail -4; done
