"""Convert the reference's Schaefer-Turek dolfin-XML meshes to `.npz` fixtures.

Run once in the build container (the reference tree does not exist on the GPU
box):  python tests/golden/convert_meshes.py

Source data: /root/reference/tests/mesh/cylinder_{0,1,2,3}.xml  (data files the
reference's own drivers load, `problem_setups.py:476-478`).  Output: vertex
coordinates + triangle connectivity only.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from dolfin_navier_scipy_amd.fem.mesh2d import read_dolfin_xml, save_npz_mesh  # noqa

REFMESH = '/root/reference/tests/mesh'

if __name__ == '__main__':
    for lvl in (0, 1, 2, 3):
        mesh = read_dolfin_xml(os.path.join(REFMESH,
                                            'cylinder_{0}.xml'.format(lvl)))
        out = os.path.join(HERE, 'mesh_cylinder_{0}.npz'.format(lvl))
        save_npz_mesh(out, mesh)
        print(out, mesh.nverts, mesh.ncells)
