"""In-process power / clock / energy readings through ``librocm_smi64`` (ctypes, no subprocess).

Evidence only (the ``power`` object of ``bench.py``, ``tools/energy_*.py``); nothing on the hot path
uses it.  The library reads sysfs and never creates a HIP context, so loading it in a process that
runs kernels is harmless - unlike forking the ``rocm-smi`` script from such a process, which under
``rocprofv3 --pmc`` is the exec the GPU pool forbids.

    with Smi(device=0) as s:          # raises SmiError when the library or the device is missing
        j0 = s.energy_j()
        ...
        watts, mhz = s.power_w(), s.sclk_mhz()
"""
from __future__ import annotations

import ctypes
import os

_MAX_FREQ = 33          # RSMI_MAX_NUM_FREQUENCIES (rocm_smi.h)
_CLK_SYS = 0            # RSMI_CLK_TYPE_SYS


class SmiError(RuntimeError):
    pass


class _Frequencies(ctypes.Structure):
    # rsmi_frequencies_t of ROCm >= 6 (this image: 7.2): the leading has_deep_sleep flag is part of the layout
    _fields_ = [("has_deep_sleep", ctypes.c_bool), ("num_supported", ctypes.c_uint32),
                ("current", ctypes.c_uint32), ("frequency", ctypes.c_uint64 * _MAX_FREQ)]


def _load():
    names = []
    if os.environ.get("PF_SMI_LIB"):
        names.append(os.environ["PF_SMI_LIB"])
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    names += [os.path.join(rocm, "lib", "librocm_smi64.so.1"), "/opt/rocm/lib/librocm_smi64.so.1",
              "librocm_smi64.so.1", "librocm_smi64.so"]
    last = None
    for n in names:
        try:
            return ctypes.CDLL(n)
        except OSError as exc:
            last = exc
    raise SmiError(f"cannot load librocm_smi64 ({last})")


class Smi:
    """One device of ``librocm_smi64``.  ``device`` is the library's own index (PCI order)."""

    def __init__(self, device: int = 0, pci: "tuple[int, int, int] | None" = None):
        """``pci`` = (domain, bus, device) of the GPU to watch - HIP ordinals follow ``HIP_VISIBLE_DEVICES``,
        rocm_smi indexes in PCI order, so a caller that knows its engine's bus address passes it and ``device``
        is ignored (``self.index`` / ``self.bdf`` say what was found)."""
        self._open = False
        self._lib = _load()
        self._dev = ctypes.c_uint32(device)
        rc = self._lib.rsmi_init(ctypes.c_uint64(0))
        if rc != 0:
            raise SmiError(f"rsmi_init failed ({rc})")
        self._open = True
        n = ctypes.c_uint32(0)
        if self._lib.rsmi_num_monitor_devices(ctypes.byref(n)) != 0:
            self.close()
            raise SmiError("rsmi_num_monitor_devices failed")
        if pci is not None:
            try:
                found = [i for i in range(n.value) if self._bdf(i)[:3] == tuple(pci)]
            except SmiError:
                self.close()
                raise
            if not found:
                self.close()
                raise SmiError(f"rocm_smi has no device at PCI {pci[0]:04x}:{pci[1]:02x}:{pci[2]:02x}")
            device = found[0]
            self._dev = ctypes.c_uint32(device)
        if device >= n.value:
            self.close()
            raise SmiError(f"rocm_smi sees {n.value} device(s), wanted index {device}")
        self.index = device
        try:
            d, b, dv, fn = self._bdf(device)
            self.bdf = f"{d:04x}:{b:02x}:{dv:02x}.{fn:x}"
        except SmiError:
            self.bdf = None

    def _bdf(self, index: int):
        """(domain, bus, device, function) of rocm_smi device ``index`` (rsmi_dev_pci_id_get: domain in bits
        63..32, bus 15..8, device 7..3, function 2..0; bits 31..28 carry a partition id on newer releases)."""
        v = ctypes.c_uint64(0)
        if self._lib.rsmi_dev_pci_id_get(ctypes.c_uint32(index), ctypes.byref(v)) != 0:
            raise SmiError("rsmi_dev_pci_id_get failed")
        x = v.value
        return (x >> 32) & 0xffffffff, (x >> 8) & 0xff, (x >> 3) & 0x1f, x & 0x7

    def close(self):
        if getattr(self, "_open", False):
            self._lib.rsmi_shut_down()
            self._open = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def energy_j(self) -> float:
        """Accumulated socket energy in joules (the counter ticks in units of ~15.3 uJ)."""
        cnt, res, ts = ctypes.c_uint64(0), ctypes.c_float(0), ctypes.c_uint64(0)
        rc = self._lib.rsmi_dev_energy_count_get(self._dev, ctypes.byref(cnt), ctypes.byref(res), ctypes.byref(ts))
        if rc != 0:
            raise SmiError(f"rsmi_dev_energy_count_get failed ({rc})")
        return cnt.value * float(res.value) * 1e-6

    def power_w(self) -> float:
        """Current socket power in watts (falls back to the averaged sensor)."""
        uw = ctypes.c_uint64(0)
        if self._lib.rsmi_dev_current_socket_power_get(self._dev, ctypes.byref(uw)) == 0 and uw.value:
            return uw.value * 1e-6
        if self._lib.rsmi_dev_power_ave_get(self._dev, ctypes.c_uint32(0), ctypes.byref(uw)) == 0:
            return uw.value * 1e-6
        raise SmiError("no power sensor")

    def sclk_mhz(self) -> float:
        f = _Frequencies()
        rc = self._lib.rsmi_dev_gpu_clk_freq_get(self._dev, ctypes.c_int(_CLK_SYS), ctypes.byref(f))
        if rc != 0 or f.current >= _MAX_FREQ:
            raise SmiError(f"rsmi_dev_gpu_clk_freq_get failed ({rc})")
        return f.frequency[f.current] * 1e-6
