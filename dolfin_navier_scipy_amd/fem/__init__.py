"""Host-side Taylor-Hood input scaffolding (no FEniCS on the build/GPU boxes)."""
from .mesh2d import (Mesh2D, read_dolfin_xml, load_npz_mesh, save_npz_mesh,
                     refine_uniform, rectangle_mesh, channel_cylinder_mesh)
from .taylor_hood import TaylorHood
from .problem_setups import (get_sysmats, condense_sysmatsbybcs,
                             cylinder_mesh, cylinder_mesh_hierarchy,
                             pressure_prolongations, DATA_DIR,
                             gen_bccont_fems, classify_boundary)
