"""MI355X-native saddle-point time stepping for dolfin_navier_scipy.

One hot path, re-built for gfx950: the per-time-step solve
`[[M + theta*dt*(A + N), J^T], [J, 0]] [v; p] = rhs` of
`stokes_navier_utils.solve_nse` / `time_int_utils.cnab|sbdftwo`, behind the
reference's own linear-algebra boundary (`lin_alg_utils`).

 * `lin_alg_utils`  -- drop-in for `sadptprj_riclyap_adi.lin_alg_utils`
 * `time_int_utils` -- drop-in for the reference's semi-explicit integrators
 * `saddle`         -- handles on the HBM-resident system / stepper (C-ABI)
 * `fem`            -- host-side Taylor-Hood scaffolding that produces inputs
"""
import sys
import types

__version__ = '0.1.0'


def install_as_lau():
    """make `import sadptprj_riclyap_adi.lin_alg_utils as lau` (reference
    tiu:9, snu:291,723,1614) resolve to the MI355X implementation"""
    from . import lin_alg_utils
    pkg = sys.modules.get('sadptprj_riclyap_adi')
    if pkg is None:
        pkg = types.ModuleType('sadptprj_riclyap_adi')
        pkg.__path__ = []
        sys.modules['sadptprj_riclyap_adi'] = pkg
    pkg.lin_alg_utils = lin_alg_utils
    sys.modules['sadptprj_riclyap_adi.lin_alg_utils'] = lin_alg_utils
    return lin_alg_utils
