timeout -k 5 300 python tools/r5_c4.py cube panini "f_fov 180" 2>&1 | grep C4PROBE | head -3
timeout -k 5 300 python tools/r5_c4.py cube hammer "" 2>&1 | grep C4PROBE | head -3
