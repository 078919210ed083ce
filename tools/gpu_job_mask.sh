#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/mask
timeout 1200 python -m pytest tests/test_fast_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "mask or relu" 2>&1 | tail -6 | tee gpurun_out/mask/pytest.txt
timeout 300 python tools/k4_mask_time.py 2>&1 | tee gpurun_out/mask/time.txt
