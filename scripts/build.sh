#!/bin/bash
# build the gfx950 library; non-zero exit on failure
cd "$(dirname "$0")/.." && python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -E "error|built" -A4 | head -30
python - <<'PY'
import os, sys
sys.path.insert(0, '.')
from dolfin_navier_scipy_amd.build import needs_build
sys.exit(1 if needs_build() else 0)
PY
