#!/bin/bash
# like bench_matrix.sh, with the CPU leg (parity) kept
for a in "$@"; do
  timeout -k 10 300 python bench.py --roofline-refine 0 --no-picard --no-refined $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$a', '| steps/s %.0f its %.2f relres %.2e parity v %.2e p %.2e' % (d['value'], c['krylov_iters_per_step'], c['true_relres_last'], d['parity']['v_rel_Mnorm'], d['parity']['p_rel_l2']))"
done
