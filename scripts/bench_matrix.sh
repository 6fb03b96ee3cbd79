#!/bin/bash
# experiment: launch shapes of the search bench (batch x streams x resident-wave cap), one line per configuration
run() { tag=$1; shift; python bench.py --no-cpu-baseline --no-clustered --no-extras "$@" > gpurun_out/bm_$tag.log 2>&1; python scripts/show_bf16.py gpurun_out/bm_$tag.log; }
C3="--nodes 1000000 --dim 768 --m 32 --ef 400 --k 100 --batch 4096 --graph fast --steps 30 --warmup 3"
run c3_s1 $C3 --streams 1
run c3_s2 $C3 --streams 2
run c3_s3 $C3 --streams 3
C4="--nodes 10000000 --graph fast --steps 60"
run c4_s2 $C4 --streams 2
run c4_s3 $C4 --streams 3
run c4_s4 $C4 --streams 4
