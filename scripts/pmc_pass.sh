#!/bin/bash
# usage: scripts/pmc_pass.sh <tag> <bench args...>   -- four counter passes of bench.py under rocprofv3 (GPU box only).
# The four TCC counters do not fit one pass (the run stalls): FETCH_SIZE/TCC_HIT and WRITE_SIZE/TCC_MISS are collected separately.
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES -d $out/p1 -o p1 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > $out.p1.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA -d $out/p2 -o p2 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > $out.p2.log 2>&1 &&
echo "pass 3" &&
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum -d $out/p3 -o p3 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > $out.p3.log 2>&1 &&
echo "pass 4" &&
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum -d $out/p4 -o p4 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras "$@" > $out.p4.log 2>&1
