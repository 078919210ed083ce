#!/bin/bash
# round 5, first job: the producer's gated rescaling pass + the new sample rows, then the whole GPU suite and a short bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_producer_gpu.py -q -m gpu --tb=short -x 2>&1 | grep -v amdgpu.ids | tail -40 > gpurun_out/r5a_producer.txt
timeout 2400 python -m pytest tests -q -m gpu --tb=short 2>&1 | grep -v amdgpu.ids | tail -60 > gpurun_out/r5a_tests.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r5a_bench.json 2> gpurun_out/r5a_bench.err
tail -30 gpurun_out/r5a_producer.txt; tail -30 gpurun_out/r5a_tests.txt; tail -c 600 gpurun_out/r5a_bench.err; tail -c 1500 gpurun_out/r5a_bench.json
