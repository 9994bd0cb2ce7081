#!/bin/bash
# Kernel times of alternative builds of libseeksv_hip.so side by side: tools/gpu_variants.sh "<bench args>" <kernel regex> dir1 dir2 ...   (each dir = an SSV_LIBDIR)
args=$1; re=$2; shift 2
for d in "$@"; do
  export SSV_LIBDIR=$GRAFT_REPO_ROOT/$d
  KSTATS_LINES=40 bash $GRAFT_REPO_ROOT/tools/gpu_kstats.sh var_$(basename $d) $args | grep -E "$re" | sed "s|^|$(basename $d)  |"
done
