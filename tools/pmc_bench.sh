#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, one counter per pass) of the flow kernels inside the default bench.py run itself.
out=gpurun_out/${1:-r03q}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $out/bench_$c.log 2>&1 || exit 1
    python tools/pmc_by_kernel.py $out/pmc_$c est_tail est_resnet attn_flash attn_relpos gemm_win skinny3 decode_attn dac_ru > $out/pmc_bench_$c.txt 2>&1
    rm -rf $out/pmc_$c
    echo "pmc bench $c done" >> $out/progress.log
done
cat $out/pmc_bench_*.txt
