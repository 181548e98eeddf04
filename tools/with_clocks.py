"""Run a command while sampling the GPU's shader clock and socket power (maskbit_amd/telemetry.py); print the summary after it.
usage: python tools/with_clocks.py <command ...>"""
import json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskbit_amd.telemetry import ClockSampler

with ClockSampler(0, period_s=0.05) as cs:
    rc = subprocess.call(sys.argv[1:])
print("telemetry:", json.dumps(cs.summary()))
sys.exit(rc)
