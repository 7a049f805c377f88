#!/bin/bash
# same-box A/B of library builds: bench line (value, fused ms, tree ms) per build, order given by the arguments
for tag in "$@"; do
  if [ "$tag" = head ]; then unset AZ_MCTS_LIB; else export AZ_MCTS_LIB=$PWD/tools/probes/libaz_mcts_$tag.so; fi
  timeout -k 10 100 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$tag',d['value'],d['roofline']['ms_per_launch'],d['roofline_tree_kernel']['ms_per_launch'])" || exit 1
done
