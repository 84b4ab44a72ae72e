#!/usr/bin/python3
"""Drop-in `refine3d` for PYP, the full implementation (bin/refine3d, the compiled front end of the default call, hands everything else here; point frealign_paths["cistem2"] at this directory)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pyp_amd.surface import warm  # noqa: E402

warm.start(mb=64)      # GPU context creation overlaps the imports and the input parsing
from pyp_amd.surface import cli  # noqa: E402

sys.exit(cli.refine3d_main())
