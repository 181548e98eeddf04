"""Shader clock and socket power of the GPU while a benchmark's timed region runs.

MI355X is power-capped on this workload: with all 256 CUs multiplying random fp16 data the shader clock settles at 1.9-2.0 GHz instead of the
2.4 GHz the nominal MFMA peak assumes, and boxes differ by a few percent (profiles/r03_power_and_streams.md).  ``bench.py`` therefore samples the
clock on a side thread and prints it next to every kernel fraction, so that two runs that disagree can be told apart: a slower kernel or a slower box.

Source, in order of preference: the amdgpu hwmon files of the device (``freq1_input`` = sclk in Hz, ``power1_average`` / ``power1_input`` in uW: a
file read, no subprocess), else ``rocm-smi --showclocks --showpower --json`` once a second.  Measurement only: nothing in the product path reads it.
"""
from __future__ import annotations

import glob
import json
import os
import statistics
import subprocess
import threading
import time
from typing import List, Optional, Tuple


def _hwmon_dir(pci_bus_id: Optional[str]) -> Optional[str]:
    cands = []
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
        for hw in glob.glob(os.path.join(card, "device", "hwmon", "hwmon*")):
            if os.path.exists(os.path.join(hw, "freq1_input")):
                try:
                    bdf = os.path.basename(os.path.realpath(os.path.join(card, "device")))
                except OSError:
                    bdf = ""
                cands.append((bdf, hw))
    if not cands:
        return None
    if pci_bus_id:
        for bdf, hw in cands:
            if bdf.lower().endswith(pci_bus_id.lower()) or pci_bus_id.lower().endswith(bdf.lower()):
                return hw
    return cands[0][1] if len(cands) == 1 else None      # several cards and no match: do not guess


def _read_int(path: str) -> Optional[int]:
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


def _smi_sample(device_index: int = 0) -> Tuple[Optional[float], Optional[float]]:
    """One rocm-smi reading of THIS device (`-d index`; the entry keyed `card<index>` when several come back) -- never another card's."""
    try:
        out = subprocess.run(["rocm-smi", "-d", str(device_index), "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
        cards = json.loads(out)
        card = cards.get(f"card{device_index}")
        if card is None:
            if len(cards) != 1:
                return None, None                                  # several cards and none is named as ours: report nothing rather than GPU 0's
            card = next(iter(cards.values()))
        sclk = next((v for k, v in card.items() if "sclk" in k.lower()), None)
        pw = next((v for k, v in card.items() if "power" in k.lower() and "(w)" in k.lower()), None)
        mhz = float("".join(ch for ch in str(sclk) if ch.isdigit() or ch == ".")) if sclk is not None else None
        return mhz, float(pw) if pw is not None else None
    except Exception:      # noqa: BLE001  (telemetry never costs the benchmark)
        return None, None


class ClockSampler:
    """``with ClockSampler(device_index) as cs: <timed region>`` then ``cs.summary()``."""

    def __init__(self, device_index: int = 0, period_s: float = 0.1):
        self.period = period_s
        self.samples: List[Tuple[Optional[float], Optional[float]]] = []
        self._stop = threading.Event()
        self._thread: Optional[threading.Thread] = None
        bdf = None
        try:
            import torch
            p = torch.cuda.get_device_properties(device_index)
            if hasattr(p, "pci_bus_id"):
                bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{getattr(p, 'pci_device_id', 0):02x}.0"
        except Exception:      # noqa: BLE001
            pass
        self.device_index = device_index
        self.hwmon = _hwmon_dir(bdf)
        self.source = f"hwmon ({self.hwmon})" if self.hwmon else f"rocm-smi -d {device_index}"
        if not self.hwmon:
            self.period = max(self.period, 1.0)

    def _one(self):
        if self.hwmon:
            hz = _read_int(os.path.join(self.hwmon, "freq1_input"))
            uw = _read_int(os.path.join(self.hwmon, "power1_average"))
            if uw is None:
                uw = _read_int(os.path.join(self.hwmon, "power1_input"))
            return (hz / 1e6 if hz else None, uw / 1e6 if uw else None)
        return _smi_sample(self.device_index)

    def _run(self):
        while not self._stop.is_set():
            self.samples.append(self._one())
            self._stop.wait(self.period)

    def __enter__(self):
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=10)
        return False

    def summary(self) -> dict:
        clk = [c for c, _ in self.samples if c]
        pw = [p for _, p in self.samples if p]
        out = {"source": self.source, "samples": len(self.samples)}
        if clk:
            out.update(effective_clock_mhz=statistics.median(clk), clock_mhz_min=min(clk), clock_mhz_max=max(clk))
        if pw:
            out.update(socket_power_w=statistics.median(pw), socket_power_w_max=max(pw))
        return out
