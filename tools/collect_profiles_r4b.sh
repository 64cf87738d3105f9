#!/bin/bash
# Second collection of round 4 (after the 64-row split tile, the 8-wave est_resnet / decode attention): kernel stats of the default
# bench (split build), the bf16 build, the fp32-kind checkpoint, the decode loop alone, the isolated CFM solve; PMC passes over the
# CFM solve (FETCH / WRITE / SQ) and over the single-thread bench step.  usage: tools/collect_profiles_r4b.sh <tag>
out=gpurun_out/${1:-r4bp}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
kstats() {      # name, command...
    local name=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$name -- "$@" > $out/$name.log 2>&1
    find $out/kt_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${name}_kernel_stats.csv
    rm -rf $out/kt_$name
    echo "kernel stats $name done" >> $out/progress.log
}
pmc() {         # name, counters, filter words, command...
    local name=$1 ctr=$2 keep=$3; shift 3
    rocprofv3 --pmc $ctr --output-format csv -d $out/pmc_$name -- "$@" > $out/pmc_$name.log 2>&1
    python tools/pmc_by_kernel.py $out/pmc_$name $keep > $out/pmc_$name.txt 2>&1
    [ -s $out/pmc_$name.txt ] || { echo "no counters for $name:"; tail -5 $out/pmc_$name.log; }
    rm -rf $out/pmc_$name
    echo "pmc $name done" >> $out/progress.log
}
kstats bench_x python3 bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline &&
kstats bench_bf16 python3 bench.py --dtype bf16 --steps 5 --warmup 2 --no-extras --no-cpu-baseline &&
kstats bench_x_fp32ckpt python3 bench.py --checkpoint fp32 --steps 3 --warmup 1 --no-extras --no-cpu-baseline &&
kstats decode_x python3 tools/decode_alone.py --dtype x --steps 2 &&
kstats cfm8x896_x python3 tools/prof_cfm.py 8 896 x || exit 1
for c in "FETCH_SIZE" "WRITE_SIZE"; do
    pmc cfmx_$c "$c" "est_tail est_resnet attn_flash gemm_win rownorm" python3 tools/prof_cfm.py 8 896 x || exit 1
done
pmc cfmx_SQ "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "est_tail est_resnet attn_flash gemm_win" python3 tools/prof_cfm.py 8 896 x || exit 1
bash tools/pmc_bench.sh ${1:-r4bp} x
ls $out
