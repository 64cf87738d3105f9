#!/bin/bash
# Scheduling sweep of the config-4 rank share (bench.py flags), one line per setting: value, ms per step, LM loop / tail times.
#   bash tools/sweep_sched.sh OUTDIR [dtype]
out=$1; dt=${2:-bf16}; mkdir -p $out
run() {
  name=$1; shift
  MMX_TIMING=1 python bench.py --dtype $dt --steps 3 --warmup 2 --no-cpu-baseline --no-extras "$@" > $out/$name.json 2> $out/$name.err
  v=$(python -c "import json,sys; d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  t=$(grep "LM loop done" $out/$name.err $out/$name.json | tail -1 | sed 's/.*LM loop done at/LM/; s/, decode steps.*//')
  echo "$name: $v | $t" | tee -a $out/sweep.txt
}
if [ "$3" = "w1" ]; then
  run base
  run workers1 --flow-workers 1
  run workers1_np --flow-workers 1 --no-polite
  run workers1_h30 --flow-workers 1 --hold-steps 30
  exit 0
fi
if [ "$3" = "groups" ]; then
  run base
  run pad15 --pad-ratio 1.5
  run pad30 --pad-ratio 3.0
  run ramp248 --flow-group 2,4,8
  run ramp48 --flow-group 4,8
  run hold50 --hold-steps 50
  run hold70 --hold-steps 70
  exit 0
fi
if [ "$3" = "prio" ]; then
  run base
  run prio1 --flow-priority 1
  run prio2 --flow-priority 2
  exit 0
fi
if [ "$3" = "lm" ]; then
  run base
  run down24 --lm-cfg down=2,4
  run down22 --lm-cfg down=2,2
  run gu11 --lm-cfg gu=1,1
  run down18 --lm-cfg down=1,8
  exit 0
fi
if [ "$3" = "fan" ]; then
  run fan3
  run fan1 --group-fan 1
  run fan2 --group-fan 2
  run fan4 --group-fan 4
  exit 0
fi
run base
run nopolite --no-polite
run hold40 --hold-steps 40
run hold90 --hold-steps 90
run tail3 --tail-active 3
run group6 --flow-group 6
run group12 --flow-group 12
run workers3 --flow-workers 3
run poll4 --poll-every 4
