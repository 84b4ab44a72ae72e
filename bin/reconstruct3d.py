#!/usr/bin/python3
"""Drop-in `reconstruct3d` for PYP (point frealign_paths["cistem2"] at this directory)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pyp_amd.surface import warm  # noqa: E402

warm.start()      # GPU context creation overlaps the imports and the input parsing
from pyp_amd.surface import cli  # noqa: E402

sys.exit(cli.reconstruct3d_main())
