#!/bin/bash
# usage (GPU box): scripts/pmc_custom.sh <tag> "<bench args>" "<counters pass 1>" ["<counters pass 2>" ...]
# one rocprofv3 --pmc pass of bench.py (3 steps) per counter list, each under its own timeout; CSVs under gpurun_out/pmc_<tag>/pN
tag=$1; bargs=$2; shift 2
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp
k=0
for counters in "$@"; do
  k=$((k+1))
  timeout -k 10 100 rocprofv3 --pmc $counters -d $out/p$k -o p$k --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras $bargs > $out.p$k.log 2>&1 || { echo "pass $k failed or timed out: $counters"; exit 1; }
done
