#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/split
ILL=1 timeout 900 python tools/split_ab.py 12 2>&1 | tee gpurun_out/split/ab_ill.txt
