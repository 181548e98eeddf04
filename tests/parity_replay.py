"""The replay of the reference's recorded runs lives in the package (maskbit_amd/parity_replay.py: bench.py and the tools use it too)."""
from maskbit_amd.parity_replay import *  # noqa: F401,F403
from maskbit_amd.parity_replay import GOLDEN, GOLDEN_DIR, RUN_C3_S2, RUN_C3_S3, RUN_CFG1, RUN_CFG1_S2, RUN_CFG5, RUN_CFG5_S2  # noqa: F401
