#!/bin/bash
# rocprofv3 kernel stats of leandvb_amd in steady state (capture ×10): tools/app_prof.sh [app flags…] → gpurun_out/app_stats.csv
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd "$REPO"
python tools/chain_bench.py 8000 "$@" --repeat 10 > /dev/null 2>&1   # builds /tmp/cap_8000x10.u8
export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/app_prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/app_prof -- "$REPO/leansdr_amd/host/apps/leandvb_amd" --u8 -f 2400e3 --sr 2000e3 --cr 1/2 "$@" < /tmp/cap_8000x10.u8 > /dev/null 2> /tmp/app_prof.log
f=$(find /tmp/app_prof -name "*kernel_stats.csv" | head -1)
mkdir -p "$REPO/gpurun_out"; cp "$f" "$REPO/gpurun_out/app_stats.csv"
head -12 "$f" | cut -c1-160
