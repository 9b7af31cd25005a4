mkdir -p gpurun_out/r04
timeout 300 python tools/probe/f16_probe.py > gpurun_out/r04/f16_probe.txt 2>&1
timeout 200 python tools/exp_power.py > gpurun_out/r04/power.txt 2>&1
timeout 200 python tools/exp_launch_length.py > gpurun_out/r04/launch_length.txt 2>&1
timeout 200 python tools/probe/mfma_probe.py > gpurun_out/r04/mfma_probe.txt 2>&1
tail -50 gpurun_out/r04/f16_probe.txt
