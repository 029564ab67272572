#!/bin/bash
# one GPU round of the headline kernel: parity of the batched path, then timing + phase profile (gpurun_out/quick.log)
timeout -k 10 300 python -m pytest tests/test_gpu_batch.py tests/test_gpu_chain_modes.py -x -q > gpurun_out/gputest.log 2>&1; echo rc=$? >> gpurun_out/gputest.log; tail -4 gpurun_out/gputest.log
timeout -k 10 120 python tools/quick_time.py > gpurun_out/quick.log 2>&1; grep -v "^gen\|amdgpu.ids\|^lds" gpurun_out/quick.log
