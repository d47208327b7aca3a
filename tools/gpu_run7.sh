set -x
mkdir -p gpurun_out
timeout 600 python tools/sweep.py lds_pad_bytes 0,8192,18000,32000,60000,100000 > gpurun_out/sweep_occ.log 2>&1
bash tools/gpu_pmc.sh v3 > gpurun_out/pmc_v3.log 2>&1
echo done
