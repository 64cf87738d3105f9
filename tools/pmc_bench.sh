#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, one counter per pass, no trace domain next to --pmc) of the flow kernels inside bench.py's
# own step.  Round 3's pass over the multi-threaded step died in the profiler's counter read-back thread (profiles/README.md);
# this pass runs the step with --no-overlap (decode loop, then the flow groups, all issued by ONE host thread on one stream: the
# same kernels over group shapes of the same sizes) and keeps the process map (MMX_DUMP_MAPS) so that a fault report can be
# symbolised.  usage: tools/pmc_bench.sh <tag> [x|bf16]
out=gpurun_out/${1:-r04q}
dt=${2:-x}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
    MMX_DUMP_MAPS=$GRAFT_REPO_ROOT/$out/maps_$c.txt rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py --dtype $dt --steps 1 --warmup 0 --no-overlap --no-roofline --no-cpu-baseline --no-extras > $out/bench_$c.log 2>&1 || { echo "pmc bench $c FAILED" >> $out/progress.log; tail -20 $out/bench_$c.log; exit 1; }
    python tools/pmc_by_kernel.py $out/pmc_$c est_tail est_resnet attn_flash attn_relpos gemm_win skinny3 decode_attn dac_ru sample_step > $out/pmc_bench_${dt}_$c.txt 2>&1
    rm -rf $out/pmc_$c
    echo "pmc bench $c done" >> $out/progress.log
done
cat $out/pmc_bench_${dt}_*.txt
