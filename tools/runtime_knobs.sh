#!/bin/bash
# dependent-kernel boundary cost under graph replay (tools/bench_launch.py) for the HIP runtime's own knobs: kernel arguments in device
# memory, pre-captured AQL packets, fence scopes.  usage (GPU box, repo root): bash tools/runtime_knobs.sh [bench]
run() { echo "== $*"; env "$@" python tools/bench_launch.py 2>&1 | grep -E "trivial|resadd_ln 14|skinny 512x2048 \(2 MB|tiled 128x512x64"; }
run A=0
run HIP_FORCE_DEV_KERNARG=0
run HIP_FORCE_DEV_KERNARG=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run AMD_OPT_FLUSH=0
run ROC_SYSTEM_SCOPE_SIGNAL=0
run ROC_USE_FGS_KERNARG=0
run DEBUG_HIP_KERNARG_COPY_OPT=0
run DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1
run GPU_FLUSH_ON_EXECUTION=1
run HSA_ENABLE_SDMA=0
run HSA_ENABLE_INTERRUPT=0
