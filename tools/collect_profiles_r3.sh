#!/bin/bash
# Round-3 profile collection on the GPU box.  Kernel stats (rocprofv3 --kernel-trace --stats) of the default bench (bf16), of
# the split build's bench, of the decode loop alone (both builds), of an isolated batched CFM solve (both builds) and of the DAC
# decoder; PMC passes (one counter set per pass, no trace domains next to --pmc) over eager decode steps (HBM bytes of the
# projection / attention kernels; counters are not collected for kernels replayed from a hipGraph), over the CFM solve (HBM
# bytes, MFMA busy, LDS conflicts of the flow kernels) and over the DAC decoder (HBM bytes).  Output under gpurun_out/$1; the
# summaries judged are copied into profiles/ afterwards.
out=gpurun_out/${1:-r3p}
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
    [ -s $out/pmc_$name.txt ] || { echo "no counters for $name:"; tail -5 $out/pmc_$name.log; find $out/pmc_$name -name "*.csv" | head; }
    rm -rf $out/pmc_$name
    echo "pmc $name done" >> $out/progress.log
}
if [ "$2" != "pmc" ]; then
kstats bench python3 bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline &&
kstats bench_x python3 bench.py --dtype x --steps 3 --warmup 1 --no-extras --no-cpu-baseline &&
kstats decode_bf16 python3 tools/decode_alone.py --dtype bf16 --steps 2 &&
kstats decode_x python3 tools/decode_alone.py --dtype x --steps 2 &&
kstats cfm8x896 python3 tools/prof_cfm.py 8 896 bf16 &&
kstats cfm8x896_x python3 tools/prof_cfm.py 8 896 x &&
kstats dac python3 tools/prof_dac.py || exit 1
fi
for c in "FETCH_SIZE" "WRITE_SIZE"; do
    pmc decode_bf16_$c "$c" "skinny3 decode_attn sample_step decode_prep" python3 tools/prof_decode.py bf16 8 || exit 1
    pmc decode_x_$c "$c" "skinny3 decode_attn sample_step decode_prep" python3 tools/prof_decode.py x 8 || exit 1
    pmc dac_$c "$c" "gemm_win dac_ru conv_cout1" python3 tools/prof_dac.py || exit 1
done
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16" "FETCH_SIZE" "WRITE_SIZE"; do
    n=$(echo $c | cut -d" " -f1)
    pmc cfm_$n "$c" "est_tail est_resnet attn_flash gemm_win rownorm" python3 tools/prof_cfm.py 8 896 bf16 || exit 1
    pmc cfmx_$n "$c" "est_tail est_resnet attn_flash gemm_win rownorm" python3 tools/prof_cfm.py 8 896 x || exit 1
done
ls $out
