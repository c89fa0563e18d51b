"""Experiment helper (not part of the product): HIP streams restricted to disjoint sets of compute units
(``hipExtStreamCreateWithCUMask``); used by ``bench.py --main-cus``.  Findings: HISTORY.md, round 4, "CU-masked streams".

Two kernels overlap safely on this hardware only when their waves never share a SIMD (a half-precision
MFMA wave with lane-masked code disturbs waves of OTHER kernels on its SIMD).  Streams whose CU masks are complementary cannot
co-schedule waves on one CU, whatever the kernels' register / LDS footprints are, so a main stream on most of the chip and a
side stream on the remaining CUs may run different kernels at the same time.  Device memory, events and ordering are
PyTorch's / HIP's; this module is plumbing (ctypes on the HIP runtime PyTorch already loaded).
"""
from __future__ import annotations

import ctypes as C

import torch

_hip = None
_PAIRS: dict = {}  # (device index, side CUs) -> CuStreamPair: HIP multiplexes streams onto few hardware queues, so pairs are created once


def _runtime():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipExtStreamCreateWithCUMask.restype = C.c_int
        _hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    return _hip


def side_mask_bits(n_cus: int, n_side: int, n_xcc: int = 8):
    """Mask bits of the side stream.  Measured on MI355X (tools/micro/cumask_probe.hip, profiles/r04_cumask_probe.txt): mask bit k
    names CU k // n_xcc of XCC k % n_xcc, and an XCC whose part of the mask is EMPTY is not restricted at all — so both masks must
    keep at least one CU in every XCC.  The side stream takes bits 0 .. n_side-1 (n_side / n_xcc CUs of every XCC), which also
    keeps the main stream's persistent kernels balanced over the XCDs."""
    if n_side <= 0 or n_side >= n_cus - n_xcc + 1 or n_side % n_xcc:
        raise ValueError(f"side CUs must be a positive multiple of {n_xcc} below {n_cus}")
    return list(range(n_side))


def _create(dev_index: int, bits_on, n_cus: int):
    words = (n_cus + 31) // 32
    m = (C.c_uint32 * words)()
    for b in bits_on:
        m[b >> 5] |= 1 << (b & 31)
    h = C.c_void_p()
    with torch.cuda.device(dev_index):
        rc = _runtime().hipExtStreamCreateWithCUMask(C.byref(h), words, m)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed with {rc}")
    return torch.cuda.ExternalStream(h.value, device=torch.device("cuda", dev_index))


class CuStreamPair:
    """``main`` (all CUs but the side set) and ``side`` (the side set): complementary CU masks on one device."""

    def __init__(self, device, n_side: int):
        dev = torch.device(device)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        n_cus = torch.cuda.get_device_properties(idx).multi_processor_count
        side = side_mask_bits(n_cus, n_side)
        on = set(side)
        self.n_cus, self.n_side, self.n_main = n_cus, n_side, n_cus - n_side
        self.side = _create(idx, side, n_cus)
        self.main = _create(idx, [b for b in range(n_cus) if b not in on], n_cus)


def pair(device, n_side: int) -> CuStreamPair:
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, n_side)
    if key not in _PAIRS:
        _PAIRS[key] = CuStreamPair(dev, n_side)
    return _PAIRS[key]
