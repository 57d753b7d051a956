set -x
mkdir -p gpurun_out
(timeout 60 tools/r5_probe all > gpurun_out/r05_probe.txt 2>&1; echo "rc $?" >> gpurun_out/r05_probe.txt)
(timeout 300 python tools/r5_diag.py > gpurun_out/r05_diag.txt 2>&1; echo "rc $?" >> gpurun_out/r05_diag.txt)
(timeout 600 python -m pytest tests/test_apply_resident_gpu.py -x -q -m gpu -k "4k or c5 or lensmap_is_an_error or in_memory_when or rubix_and_stripe" > gpurun_out/r05_tests1.txt 2>&1; echo "rc $?" >> gpurun_out/r05_tests1.txt)
(PASSES="FETCH_SIZE" timeout 200 tools/resident_pmc.sh gpurun_out/r05_pmc_l2.txt 1000 hammer; echo done) 
(PASSES="FETCH_SIZE" RES_ARGS="3840 2160 0 1" timeout 200 tools/resident_pmc.sh gpurun_out/r05_pmc_l2_shape1.txt 1000 hammer; echo done)
tail -3 gpurun_out/r05_tests1.txt
