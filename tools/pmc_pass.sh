#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <kernel-regex> <driver args...> ; counter groups read from stdin, one per line
# One rocprofv3 --pmc pass per group (kernel-trace only), outputs under gpurun_out/<tag>_<n>/
tag=$1; regex=$2; shift 2
export TMPDIR=/tmp
n=0
while read -r group; do
  [ -z "$group" ] && continue; case "$group" in \#*) continue;; esac
  n=$((n+1))
  out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_$n
  rm -rf "$out"; mkdir -p "$out"
  echo "$group" > "$out/group.txt"
  ( cd /tmp && timeout -k 10 90 rocprofv3 --pmc $group --kernel-include-regex "$regex" --kernel-trace -d "$out" -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/prof_driver.py "$@" ) > "$out/log.txt" 2>&1
  rc=$?
  echo "pass $n [$group] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out, stopping"; exit 1; fi
done
exit 0
