#!/bin/bash
# same-box A/B of library variants on the resident path: tools/ab_variants.sh "base b8" "31:16569,21:16569"   (alternates, three rounds)
R=${GRAFT_REPO_ROOT:-.}; cd $R
for rep in 1 2 3; do for v in $1; do
  echo -n "$v rep $rep: "; MITOFILTER_LIB=$R/mitoflex_amd/csrc/build/variants/libmitofilter_hip_$v.so python3 tools/ab_passes.py "$2" $3
done; done
