"""Convert the reference's dolfin-XML meshes to the `.npz` fixtures of
`dolfin_navier_scipy_amd/fem/data/` and copy the geometry descriptions of the
boundary-control set-ups next to them.

Run once in the build container (the reference tree does not exist on the GPU
box):  python scripts/convert_meshes.py

Source data: `/root/reference/tests/mesh/cylinder_{0,1,2,3}.xml`,
`2D-double-rotcyl_lvl{1,2}.xml.gz`, `karman2D-{rotcyl,outlets}_lvl1.xml.gz`,
`karman2D-rotcyl_lvl{3,4}.xml.gz` and the
`*_geo_cntrlbc*.json` geometry descriptions next to them
(data files the reference's own drivers load, `problem_setups.py:476-478`,
`tests/time_dep_nse_double_rotcyl_bcrob.py:7,16-18`).  Output: vertex
coordinates + triangle connectivity only; the `*_facet_region` files are NOT
needed -- their edge numbering is dolfin's, the boundary parts are classified
geometrically from the JSON instead (`fem.problem_setups.gen_bccont_fems`).
"""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from dolfin_navier_scipy_amd.fem.mesh2d import read_dolfin_xml, save_npz_mesh  # noqa

REFMESH = '/root/reference/tests/mesh'
DATA = os.path.join(ROOT, 'dolfin_navier_scipy_amd', 'fem', 'data')

if __name__ == '__main__':
    for lvl in (0, 1, 2, 3):
        mesh = read_dolfin_xml(os.path.join(REFMESH,
                                            'cylinder_{0}.xml'.format(lvl)))
        out = os.path.join(DATA, 'mesh_cylinder_{0}.npz'.format(lvl))
        save_npz_mesh(out, mesh)
        print(out, mesh.nverts, mesh.ncells)
    for lvl in (1, 2):
        mesh = read_dolfin_xml(os.path.join(
            REFMESH, '2D-double-rotcyl_lvl{0}.xml.gz'.format(lvl)))
        out = os.path.join(DATA, 'mesh_2D-double-rotcyl_lvl{0}.npz'.format(lvl))
        save_npz_mesh(out, mesh)
        print(out, mesh.nverts, mesh.ncells)
    # the meshes of the reference's known-answer tests (test_units_residuals.py:
    # 17-26, tdp_convcheck.py:85-95)
    # ... and the finer levels of the rotating-cylinder mesh (lvl4: NV = 133 334,
    # NP = 17 135): pressure spaces beyond the dense Schur block that are NOT
    # refinements of anything this repository built -- the algebraic
    # coarsening of the multigrid Schur block is tested on them
    for name in ('karman2D-rotcyl_lvl1', 'karman2D-outlets_lvl1',
                 'karman2D-rotcyl_lvl3', 'karman2D-rotcyl_lvl4'):
        mesh = read_dolfin_xml(os.path.join(REFMESH, name + '.xml.gz'))
        out = os.path.join(DATA, 'mesh_{0}.npz'.format(name))
        save_npz_mesh(out, mesh)
        print(out, mesh.nverts, mesh.ncells)
    for name in ('2D-double-rotcyl_geo_cntrlbc.json',
                 '2D-double-rotcyl_geo_cntrlbc_rotcntrl.json',
                 'karman2D-rotcyl-bm_geo_cntrlbc.json',
                 'karman2D-outlets_geo_cntrlbc.json'):
        shutil.copy(os.path.join(REFMESH, name), os.path.join(DATA, name))
        print(name)
