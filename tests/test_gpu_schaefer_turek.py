"""Physical validation of the whole path -- scaffolding assembler, device
convection, preconditioned Krylov solve, pressure scaling -- against the
Schaefer-Turek benchmark 2D-1 (steady flow around the cylinder, Re = 20; the
reference's validation target for its cylinder-wake set-up):
c_D = 5.5795, c_L = 0.010619, dp = 0.11752."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                '..', 'scripts'))


@pytest.mark.parametrize('level,nts,tol_cd,tol_dp,tol_cl',
                         [(2, 512, 0.01, 0.03, 0.25),
                          (3, 1024, 0.002, 0.002, 0.02)])
def test_drag_lift_pressure_difference(level, nts, tol_cd, tol_dp, tol_cl):
    from dolfin_navier_scipy_amd import _capi
    assert _capi.device_count() > 0
    import schaefer_turek as st
    out = st.run(N=level, refine=0, nts=nts, tend=10.0, verbose=False)
    assert out['last_change'] < 1e-7           # steady state reached
    ref = out['reference']
    assert abs(out['cD']/ref['cD'] - 1) <= tol_cd, out
    assert abs(out['dp']/ref['dp'] - 1) <= tol_dp, out
    assert abs(out['cL']/ref['cL'] - 1) <= tol_cl, out


def test_periodic_shedding_strouhal_drag_lift():
    """Schaefer-Turek 2D-2 (Re = 100, periodic): St 0.2950-0.3050,
    c_D,max 3.22-3.24, c_L,max 0.99-1.01 -- on the reference's `cylinder_3`
    mesh (NV = 19 468) the device path gives 0.3019 / 3.219 / 0.98-1.01"""
    import schaefer_turek_unsteady as stu
    out = stu.run(N=3, nts=1024, tend=18.0, verbose=False)
    assert out['periods_seen'] >= 6
    assert 0.290 <= out['St'] <= 0.310, out
    assert 3.17 <= out['cDmax'] <= 3.27, out
    assert 0.95 <= max(out['cLmax'], -out['cLmin']) <= 1.05, out
