#!/bin/bash
# SQ counters of the shading kernel for the default build and every variants/*.so (GPU box, repo root).
# usage: tools/pmc_variants.sh <workload> ...
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/pmcv
mkdir -p $O
cd /tmp
for w in "$@"; do
  for lib in default $R/variants/*.so; do
    n=$(basename $lib .so)
    if [ "$lib" != default ]; then export RTX_AMD_LIB=$lib; else unset RTX_AMD_LIB; fi
    rm -rf $O/sq_${w}_$n
    timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $O/sq_${w}_$n --output-format csv -- python3 $R/tools/kernel_times.py $w --frames 2 > $O/sq_${w}_$n.log 2>&1 || exit 1
    echo "== $w $n"; python3 $R/tools/pmc_summary.py $O/sq_${w}_$n shade | tail -1
  done
done
