#!/bin/bash
# usage (GPU box): scripts/profile_r03.sh   -- the evidence behind bench.py's roofline object for the library as built:
#   1. rocprofv3 --kernel-trace --stats of the default `python3 bench.py`            -> gpurun_out/prof_r03/stats/ (+ bench_line.json)
#   2. counter passes of the same workload (3 steps), one rocprofv3 run per list      -> gpurun_out/pmc_r03/pN/
# Counter lists respect the per-block register counts (TCC 4 raw events per pass -- FETCH_SIZE alone is 3 on gfx950, TA / TCP 2):
# a list that needs more does not fail, the profiled process stalls (profiles/r02_summary.md).  Each run has its own timeout.
# Afterwards, in the repository:  python scripts/pmc_agg.py r03 > profiles/r03_pmc_by_kernel.csv  (prefix the library's sha16 line)
export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/prof_r03
mkdir -p $out
cd /tmp
sha256sum $root/3d-point-cloud-segmentation-using-2d-img-segmentation_amd/f3d/libf3d_hip.so | cut -c1-16 > $out/lib_sha16.txt
timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o r03 -- python3 $root/bench.py > $out/bench_line.json 2> $out/bench_stderr.log || { echo "kernel-trace run failed"; tail -5 $out/bench_stderr.log; exit 1; }
tail -c 300 $out/bench_line.json; echo
bash $root/scripts/pmc_custom.sh r03 "" \
  "FETCH_SIZE TCC_HIT_sum" \
  "WRITE_SIZE TCC_MISS_sum" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD" \
  "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
