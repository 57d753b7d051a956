mkdir -p gpurun_out
timeout 300 python tools/r5_diag.py --lenses hammer,quincuncial,mercator --flags 0,32768,0,32768 > gpurun_out/r05_diag_dpp.txt 2>&1
cat gpurun_out/r05_diag_dpp.txt
