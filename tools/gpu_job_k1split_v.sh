#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/k1split
timeout 600 python tools/k1_split_variants.py 2>&1 | tee gpurun_out/k1split/variants.txt
