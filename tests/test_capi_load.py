"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/plinopt_hip.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "plinopt_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(plo_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from plinopt_amd import capi
    L = capi.lib()
    names = _declared()
    assert len(names) >= 11
    for nme in names:
        assert hasattr(L, nme), nme
    assert sorted(capi.EXPORTS) == names


def test_pack_cost_orders_like_cmpOpCount():
    from plinopt_amd import capi
    L = capi.lib()
    # default order: smaller adds+muls, then smaller adds (plinopt_optimize.h:61-63), then seed
    a = L.plo_pack_cost(10, 2, 0, 7)
    b = L.plo_pack_cost(11, 1, 0, 3)
    c = L.plo_pack_cost(9, 4, 0, 0)
    assert a < b < c
    assert L.plo_pack_cost(10, 2, 0, 7) < L.plo_pack_cost(10, 2, 0, 8)
    assert L.plo_pack_cost(5, 9, 1, 0) < L.plo_pack_cost(6, 0, 1, 0)          # OPTIMIZE_ADDITIONS
    assert (L.plo_pack_cost(5, 9, 2, 0) >> 32) == (L.plo_pack_cost(9, 5, 2, 0) >> 32)   # OPTIMIZE_SUMS


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from plinopt_amd import CSEPlan, capi
    with pytest.raises(capi.PloError) as e:
        CSEPlan(1, 1, [0, 1], [0], [1], 131071)
    assert e.value.code == capi.PLO_E_HIP


def test_argument_validation_happens_before_device_use():
    from plinopt_amd import capi
    L = capi.lib()
    csr, keep = capi.make_csr(1, 2, [0, 2], [1, 0], [1, 1])       # columns not increasing
    h = ctypes.c_void_p()
    assert L.plo_cse_plan_create(ctypes.byref(csr), 4, ctypes.byref(h)) == capi.PLO_E_ARG   # even modulus
    assert L.plo_cse_plan_create(None, 7, ctypes.byref(h)) == capi.PLO_E_ARG
