#!/bin/bash
# experiment: launch shapes of the search bench (batch x streams x resident-wave cap), one line per configuration
run() { tag=$1; shift; python bench.py --no-cpu-baseline --no-clustered --no-extras "$@" > gpurun_out/bm_$tag.log 2>&1; python scripts/show_bf16.py gpurun_out/bm_$tag.log; }
export GPU_MAX_HW_QUEUES=8
run q8_b1024_s2 --batch 1024 --streams 2
run q8_b1024_s3 --batch 1024 --streams 3
run q8_b1024_s4 --batch 1024 --streams 4
run q8_b1024_s6 --batch 1024 --streams 6
export GPU_MAX_HW_QUEUES=16
run q16_b1024_s3 --batch 1024 --streams 3
run q16_b1024_s4 --batch 1024 --streams 4
