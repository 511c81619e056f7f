#!/bin/bash
# rocprofv3 kernel-trace stats of one bench run, filtered by a kernel-name regex.  usage: tools/kstat.sh <regex> [bench args]
PAT=$1; shift
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kstat && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstat -o t -- python /root/repo/bench.py --no-cpu-baseline --no-alt-math --no-host-input --no-graph --min-seconds 0 --steps 5 --warmup 2 "$@" > /dev/null 2>&1
grep -hE "$PAT" /tmp/kstat/*kernel_stats.csv | cut -c1-220
