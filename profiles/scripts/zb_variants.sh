#!/bin/bash
# Kernel time of tile_zbuffer for every instrumented / variant build of the library present.
for so in $GRAFT_REPO_ROOT/multimodal-fusion_amd/mvfusion/libmvfusion*.so; do
  case $so in *zbdebug*|*zbspan*|*zbtiming*) continue;; esac
  echo "== $(basename $so)"
  MVFUSION_SO=$so timeout -k 10 120 python profiles/scripts/zb_phases.py ${1:-c2} | grep -v "^$" || exit 1
done
