mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r05i_tests.log
timeout 200 python __graft_entry__.py smoke > gpurun_out/r05i_smoke.log 2>&1
bash tools/profile.sh fp16x3 > gpurun_out/r05i_profile.log 2>&1
tail -4 gpurun_out/r05i_tests.log; tail -3 gpurun_out/r05i_smoke.log; tail -12 gpurun_out/r05i_profile.log
