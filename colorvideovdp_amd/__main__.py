"""`python -m colorvideovdp_amd ...` = the `cvvdp` command line (colorvideovdp_amd/cli.py)."""
import sys

from .cli import main

sys.exit(main())
