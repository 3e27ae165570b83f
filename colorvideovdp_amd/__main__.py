"""`python -m colorvideovdp_amd ...` = the `cvvdp` command line (colorvideovdp_amd/run_cvvdp.py)."""
import sys

from .run_cvvdp import main

sys.exit(main())
