timeout -k 2 100 python tools/r5_diag.py --lenses panini --flags 0,33554432,67108864,75497472 2>&1 | grep DIAG
for FL in 67117056; do
RES_DEBUG=1 timeout -k 2 60 python tools/resident_traffic.py panini 2000 3840 2160 $FL > gpurun_out/r05_v2_dbg.txt 2>&1
LAG=$(grep -o "'laggard': [0-9]*" gpurun_out/r05_v2_dbg.txt | grep -o "[0-9]*")
BLINKY_DBG_WG=$LAG RES_DEBUG=1 timeout -k 2 60 python tools/resident_traffic.py panini 2000 3840 2160 $FL > gpurun_out/r05_v2_dbg.txt 2>&1
grep -E "RESIDENT2 work" gpurun_out/r05_v2_dbg.txt | cut -c40-500
done
