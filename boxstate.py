"""What the GPU box says about itself, recorded next to every bench line (measurement tooling, not product).

Two things the judge asked to see beside a roofline fraction that moves box to box:
  * the static state: memory / compute partition, performance level, VRAM in use, driver / kernel, which of the node's
    GPUs this is (PCI address, unique id) -- `static_state()`;
  * the dynamic state DURING a timed loop: shader / memory / fabric clocks, socket power, temperatures, sampled from
    sysfs by a thread every few milliseconds -- `Sampler`.
Everything is read from /sys (the amdgpu driver's files of THIS device, found through its PCI address); nothing is
written, nothing is set.  Missing files are skipped: the object simply has fewer keys."""
from __future__ import annotations

import glob
import os
import threading
import time


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def device_sysfs(torch=None, index=0):
    """/sys/bus/pci/devices/<address> of HIP device `index` (None when it cannot be found)"""
    addr = None
    if torch is not None:
        try:
            p = torch.cuda.get_device_properties(index)
            addr = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        except Exception:  # noqa: BLE001
            addr = None
    if addr and os.path.isdir(f"/sys/bus/pci/devices/{addr}"):
        return f"/sys/bus/pci/devices/{addr}"
    cards = sorted(glob.glob("/sys/class/drm/card*/device"))
    cards = [c for c in cards if os.path.exists(os.path.join(c, "pp_dpm_sclk"))]
    return os.path.realpath(cards[index]) if len(cards) > index else None


def _active_level(text):
    """'0: 500Mhz\\n1: 2400Mhz *' -> 2400"""
    if not text:
        return None
    for line in text.splitlines():
        if line.rstrip().endswith("*"):
            tok = line.split(":")[1].strip().split("M")[0]
            try:
                return int(tok)
            except ValueError:
                return None
    return None


def static_state(torch=None, index=0):
    d = device_sysfs(torch, index)
    out = {"sysfs": d}
    if d is None:
        return out
    out["pci"] = os.path.basename(d)
    for key, name in (("memory_partition", "current_memory_partition"), ("compute_partition", "current_compute_partition"),
                      ("perf_level", "power_dpm_force_performance_level"), ("unique_id", "unique_id"),
                      ("vbios", "vbios_version")):
        v = _read(os.path.join(d, name))
        if v is not None:
            out[key] = v
    for key, name in (("vram_total", "mem_info_vram_total"), ("vram_used", "mem_info_vram_used")):
        v = _read(os.path.join(d, name))
        if v is not None:
            out[key] = int(v)
    for key, name in (("sclk_levels", "pp_dpm_sclk"), ("mclk_levels", "pp_dpm_mclk"), ("fclk_levels", "pp_dpm_fclk")):
        v = _read(os.path.join(d, name))
        if v is not None:
            out[key] = " | ".join(x.strip() for x in v.splitlines())
    hw = glob.glob(os.path.join(d, "hwmon", "hwmon*"))
    if hw:
        v = _read(os.path.join(hw[0], "power1_cap"))
        if v is not None:
            out["power_cap_w"] = int(v) / 1e6
    # the node: how many GPUs it carries and how many of them are busy (another tenant's load shares the host, the
    # chassis' power and cooling -- not the HBM)
    others = []
    for c in sorted(glob.glob("/sys/class/drm/card*/device")):
        if not os.path.exists(os.path.join(c, "pp_dpm_sclk")):
            continue
        h = glob.glob(os.path.join(c, "hwmon", "hwmon*"))
        p = _read(os.path.join(h[0], "power1_input")) if h else None
        u = _read(os.path.join(c, "mem_info_vram_used"))
        others.append(dict(pci=os.path.basename(os.path.realpath(c)), power_w=int(p) / 1e6 if p else None,
                           vram_used_gb=int(u) / 2**30 if u else None))
    out["node_gpus"] = len(others)
    out["node_gpus_busy"] = sum(1 for o in others if (o["power_w"] or 0) > 400 or (o["vram_used_gb"] or 0) > 4)
    out["node_power_w"] = sum(o["power_w"] or 0 for o in others)
    out["kernel"] = _read("/proc/sys/kernel/osrelease")
    out["amdgpu_version"] = _read("/sys/module/amdgpu/version")
    return out


class Sampler:
    """samples clocks / power / temperature of one device from sysfs while a timed loop runs:
        with Sampler(sysfs_dir) as s: ...timed loop...
        s.summary() -> {"samples": n, "sclk_mhz": {"min","mean","max"}, "power_w": {...}, ...}"""

    def __init__(self, sysfs_dir, period_s=0.004):
        self.d, self.period = sysfs_dir, period_s
        self.rows = []
        self._stop = threading.Event()
        self._t = None
        hw = glob.glob(os.path.join(sysfs_dir, "hwmon", "hwmon*")) if sysfs_dir else []
        self.hw = hw[0] if hw else None

    def _once(self):
        r = {}
        if self.hw:
            for key, name, scale in (("sclk_mhz", "freq1_input", 1e-6), ("mclk_mhz", "freq2_input", 1e-6),
                                     ("power_w", "power1_input", 1e-6), ("power_w", "power1_average", 1e-6),
                                     ("temp_hotspot_c", "temp2_input", 1e-3), ("temp_mem_c", "temp3_input", 1e-3)):
                v = _read(os.path.join(self.hw, name))
                if v is not None and key not in r:
                    try:
                        r[key] = int(v) * scale
                    except ValueError:
                        pass
        v = _active_level(_read(os.path.join(self.d, "pp_dpm_fclk"))) if self.d else None
        if v is not None:
            r["fclk_mhz"] = v
        return r

    def _run(self):
        while not self._stop.is_set():
            self.rows.append(self._once())
            self._stop.wait(self.period)

    def __enter__(self):
        if self.d:
            self._t = threading.Thread(target=self._run, daemon=True)
            self._t.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._t:
            self._t.join(timeout=1.0)
        return False

    def summary(self):
        out = {"samples": len(self.rows)}
        keys = sorted({k for r in self.rows for k in r})
        for k in keys:
            vals = [r[k] for r in self.rows if k in r]
            if vals:
                out[k] = {"min": min(vals), "mean": sum(vals) / len(vals), "max": max(vals)}
        return out


if __name__ == "__main__":
    import json
    print(json.dumps(static_state(), indent=1))
    with Sampler(device_sysfs()) as s:
        time.sleep(0.1)
    print(json.dumps(s.summary(), indent=1))
