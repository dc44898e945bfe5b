#!/bin/bash
# Degree / drop-tolerance Pareto of the explicit polynomial Fh^-1 in the
# bandwidth regime: non-zeros of Gc, Krylov steps per time step AND time steps
# per second (refined_bench.py; env MG_DEG, MG_DROP, MG_FHAT)
# usage (GPU box): bash scripts/gc_pareto.sh <tag> <refine> <nts> <nsteps> "deg:drop deg:drop ..."
TAG=${1:-r04_gc_pareto}
REF=${2:-3}
NTS=${3:-4096}
NST=${4:-100}
CONFIGS=${5:-"6:1e-3 6:3e-3 6:1e-2 8:1e-3 8:3e-3 8:1e-2"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for cfg in $CONFIGS; do
  deg=${cfg%%:*}; drop=${cfg##*:}
  f=$OUT/refine${REF}_deg${deg}_drop${drop}.json
  MG_DEG=$deg MG_DROP=$drop MG_FHAT=explicit timeout -k 10 400 python refined_bench.py $REF $NTS $NST 0 > $f 2> ${f%.json}.err || echo "failed: $cfg"
  python - <<PY
import json
try:
    d = json.loads(open('$f').read().strip().splitlines()[-1])
    print('refine {0} degree {1} drop {2:g}: nnz(Gc) {3:.1f}M = {4:.2f} x nnz(K), {5:.2f} Krylov steps per time step, {6:.0f} steps/s, set-up {7:.1f} s, step frac {8:.3f}'.format(
        d['refine'], d['cheb_degree'], d['drop_tol'], d['nnz_Gc']/1e6, d['nnz_Gc']/d['nnz_K'],
        d['krylov_iters_per_step'], d['gpu_steps_per_s'], d['setup_s'], d['roofline_step']['frac']))
except Exception as exc:
    print('no result for $cfg:', exc)
PY
done | tee $OUT/refine${REF}_table.txt
