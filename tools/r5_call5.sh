for ND in 0 1; do timeout -k 2 100 python tools/r5_lat.py panini $ND 2>&1 | grep LAT; done
timeout -k 2 100 python tools/r5_lat.py hammer 0 2>&1 | grep LAT
timeout -k 2 60 python -m pytest tests/test_abi.py -q 2>&1 | tail -2
