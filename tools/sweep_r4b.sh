mkdir -p gpurun_out/r4b_12
B="python bench.py --steps 8 --warmup 3 --no-extras --no-cpu-baseline --no-roofline"
$B > gpurun_out/r4b_12/base.json 2> gpurun_out/r4b_12/base.err
$B --tail-active 4 > gpurun_out/r4b_12/ta4.json 2> gpurun_out/r4b_12/ta4.err
$B --tail-active 8 > gpurun_out/r4b_12/ta8.json 2> gpurun_out/r4b_12/ta8.err
$B --hold-steps 40 > gpurun_out/r4b_12/hold40.json 2> gpurun_out/r4b_12/hold40.err
$B --flow-group 6 > gpurun_out/r4b_12/fg6.json 2> gpurun_out/r4b_12/fg6.err
$B --sched-adapt > gpurun_out/r4b_12/adapt.json 2> gpurun_out/r4b_12/adapt.err
$B --checkpoint fp32 > gpurun_out/r4b_12/fp32ckpt.json 2> gpurun_out/r4b_12/fp32ckpt.err
