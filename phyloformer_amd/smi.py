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

    def __init__(self, device: int = 0):
        self._lib = _load()
        self._dev = ctypes.c_uint32(device)
        rc = self._lib.rsmi_init(ctypes.c_uint64(0))
        if rc != 0:
            raise SmiError(f"rsmi_init failed ({rc})")
        self._open = True
        n = ctypes.c_uint32(0)
        if self._lib.rsmi_num_monitor_devices(ctypes.byref(n)) != 0 or device >= n.value:
            self.close()
            raise SmiError(f"rocm_smi sees {n.value} device(s), wanted index {device}")

    def close(self):
        if self._open:
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
