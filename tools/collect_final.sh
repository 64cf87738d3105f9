#!/bin/bash
# Kernel stats of the default bench command and of the long-form workload with the final code of the round, then the
# FETCH_SIZE / WRITE_SIZE passes over the bench run (tools/pmc_bench.sh).
out=gpurun_out/${1:-r02f}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in bench longform; do
    if [ $w = bench ]; then args="--steps 5 --warmup 2 --no-extras --no-cpu-baseline"; else args="--workload longform --steps 2 --warmup 1 --no-cpu-baseline --no-extras"; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$w -- python3 bench.py $args > $out/$w.log 2>&1 || exit 1
    find $out/kt_$w -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${w}_kernel_stats.csv
    rm -rf $out/kt_$w
    echo "kernel stats $w done" >> $out/progress.log
done
bash tools/pmc_bench.sh ${1:-r02f}
