#!/bin/bash
# round 2, first checkpoint: the whole GPU suite, the K1 bias/time tools, the bench line of every configuration
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s > gpurun_out/r2a_gputests.log 2>&1
python tools/k1_bias.py 128x32x32x256 > gpurun_out/r2a_k1.log 2>&1
python tools/k1_time.py >> gpurun_out/r2a_k1.log 2>&1
python bench.py --steps 10 --warmup 3 > gpurun_out/r2a_bench_cifar10_uncond.json 2> gpurun_out/r2a_bench_cifar10_uncond.err
for c in cifar10_cond stl10_uncond tinyimagenet_cond_sa; do
  python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r2a_bench_$c.json 2> gpurun_out/r2a_bench_$c.err
done
