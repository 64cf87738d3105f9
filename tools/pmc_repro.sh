#!/bin/bash
# The round-3 fault again, with the process map kept: rocprofv3 --pmc over the THREE-thread step of bench.py (decode thread + two flow
# workers queueing dispatches at once).  If the pass dies, the frames of the fault report can be put on gpurun_out/<tag>/maps_repro.txt.
# usage: tools/pmc_repro.sh <tag>      (run it LAST in a gpurun call: nothing may follow a step that died)
out=gpurun_out/${1:-r04r}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
MMX_DUMP_MAPS=$GRAFT_REPO_ROOT/$out/maps_repro.txt timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_repro -- python3 bench.py --dtype x --steps 1 --warmup 0 --no-roofline --no-cpu-baseline --no-extras > $out/repro.log 2>&1
rc=$?
echo "repro rc $rc" >> $out/progress.log
if [ $rc -eq 0 ]; then python tools/pmc_by_kernel.py $out/pmc_repro est_tail attn_flash skinny3 > $out/pmc_repro.txt 2>&1; fi
rm -rf $out/pmc_repro
tail -30 $out/repro.log | cut -c1-250
echo "repro rc $rc"
