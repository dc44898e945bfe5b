#!/bin/bash
# sweep rates (bench.py's picard_sweep_figures) with the Krylov cycle replayed
# as a graph / launched plainly and the update norm riding in the next step's
# element launch / launched on its own:  bash scripts/sweeps_modes.sh <tag>
TAG=${1:-r05_sweeps_modes}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
for g in 0 1; do for u in 0 1; do
  DNS_TRAP_GRAPH=$g DNS_TRAP_UPD_RIDE=$u python3 $R/scripts/sweep_once.py 3 1 > $OUT/graph${g}_ride${u}.json 2> $OUT/graph${g}_ride${u}.err
  echo "graph=$g ride=$u"; tail -1 $OUT/graph${g}_ride${u}.json | cut -c1-200
done; done
