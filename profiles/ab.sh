#!/bin/bash
# A/B of library variants on ONE box (boxes differ by +-10 %): alternating bench runs.
#   build here:  [AB_FLAGS=-D...] profiles/ab.sh build <name> [git-rev]   -> antsrl_amd/lib/variants/<name>.so
#   on the GPU:  profiles/ab.sh run <nameA> <nameB> [rounds] [extra bench args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$R/antsrl_amd/lib/variants
if [ "$1" = build ]; then
  mkdir -p $V; src=$R/antsrl_amd/csrc; inc=$R/include
  if [ -n "$3" ]; then
    tmp=$(mktemp -d); mkdir -p $tmp/antsrl_amd/csrc $tmp/include
    for f in $(git -C $R ls-tree --name-only $3 antsrl_amd/csrc/); do git -C $R show $3:$f > $tmp/$f; done
    git -C $R show $3:include/antsrl.h > $tmp/include/antsrl.h; src=$tmp/antsrl_amd/csrc
  fi
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -Wno-unused-function $AB_FLAGS \
    $src/*.hip -o $V/$2.so && echo built $V/$2.so
  exit $?
fi
shift; A=$1; B=$2; N=${3:-3}; shift 3
for i in $(seq $N); do for v in $A $B; do
  ANTSRL_LIB=$V/$v.so python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('%-8s ms/step=%.4f %s' % ('$v', d['ms_per_step'], d['roofline']['kernel_ms']))" || exit 1
done; done
