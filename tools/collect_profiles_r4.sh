#!/bin/bash
# Round-4 profile collection on the GPU box.  Kernel stats (rocprofv3 --kernel-trace --stats) of the default bench (split build),
# of the bf16 build's bench, of the split build on an fp32 checkpoint (weight planes), of the decode loop alone (both builds), of
# an isolated batched CFM solve (split) and of the DAC decoder; PMC passes (one counter per pass, no trace domain next to --pmc)
# over eager decode steps, the CFM solve and the single-thread bench step (tools/pmc_bench.sh).  Output under gpurun_out/$1; the
# summaries judged are copied into profiles/ afterwards.   usage: tools/collect_profiles_r4.sh <tag> [stats|pmc|all]
out=gpurun_out/${1:-r4p}
what=${2:-all}
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
if [ "$what" != "pmc" ]; then
kstats bench_x python3 bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline &&
kstats bench_bf16 python3 bench.py --dtype bf16 --steps 5 --warmup 2 --no-extras --no-cpu-baseline &&
kstats bench_x_fp32ckpt python3 bench.py --checkpoint fp32 --steps 3 --warmup 1 --no-extras --no-cpu-baseline &&
kstats decode_x python3 tools/decode_alone.py --dtype x --steps 2 &&
kstats decode_bf16 python3 tools/decode_alone.py --dtype bf16 --steps 2 &&
kstats cfm8x896_x python3 tools/prof_cfm.py 8 896 x &&
kstats dac python3 tools/prof_dac.py || exit 1
fi
if [ "$what" != "stats" ]; then
for c in "FETCH_SIZE" "WRITE_SIZE"; do
    pmc decode_x_$c "$c" "skinny3 decode_attn sample_step decode_prep" python3 tools/prof_decode.py x 8 || exit 1
    pmc decode_bf16_$c "$c" "skinny3 decode_attn sample_step decode_prep" python3 tools/prof_decode.py bf16 8 || exit 1
    pmc cfmx_$c "$c" "est_tail est_resnet attn_flash gemm_win rownorm" python3 tools/prof_cfm.py 8 896 x || exit 1
done
pmc cfmx_SQ "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "est_tail est_resnet attn_flash gemm_win" python3 tools/prof_cfm.py 8 896 x || exit 1
bash tools/pmc_bench.sh ${1:-r4p} x
fi
ls $out
