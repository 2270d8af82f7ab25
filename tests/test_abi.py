"""CPU: the C-ABI library loads and exports every symbol include/tdn.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "tdn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tdn_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    from torch_detection_amd import _lib
    names = _declared()
    assert len(names) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libtdn.so does not export %s" % n
    assert sorted(_lib.SIGNATURES) == names, set(names) ^ set(_lib.SIGNATURES)
    assert _lib.load().tdn_version() == 100


def test_epilogue_struct_layout():
    from torch_detection_amd._lib import Epilogue
    # mirror of tdn_epilogue: 3 pointers, 4 int32, pointer, 2 int32, pointer, int64 (LP64)
    assert ctypes.sizeof(Epilogue) == 72
    assert Epilogue.mask_src.offset == 40 and Epilogue.out_f32.offset == 48


def test_struct_mirrors_match_the_header(tmp_path):
    """sizeof / offsetof of every struct of include/tdn.h as gcc lays it out == the ctypes mirrors in _lib.py."""
    import subprocess
    from torch_detection_amd import _lib
    mirrors = {"tdn_epilogue": _lib.Epilogue, "tdn_wgrad_item": _lib.WgradItem, "tdn_prep_item": _lib.PrepItem}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "tdn.h"', 'int main(void) {']
    for cname, cls in mirrors.items():
        lines.append('printf("%s sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ['return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split("\n")
    seen = 0
    for ln in out:
        if not ln:
            continue
        cname, fname, val = ln.split()
        cls = mirrors[cname]
        got = ctypes.sizeof(cls) if fname == "sizeof" else getattr(cls, fname).offset
        assert got == int(val), (cname, fname, got, val)
        seen += 1
    assert seen == sum(len(c._fields_) + 1 for c in mirrors.values())


def test_host_side_plans_and_errors():
    """tdn_conv2d_plan is host-only: GEMM decomposition of the BASELINE shapes."""
    from torch_detection_amd import _lib
    lib = _lib.load()
    o = (ctypes.c_int32 * 16)()
    # neck.fpn_convs.0 forward, B=2: M=134400, N=256, K=2304
    assert lib.tdn_conv2d_plan(0, 2, 200, 336, 256, 256, 3, 1, 1, o) == 0
    assert (o[0], o[1], o[2]) == (134400, 256, 2304) and o[9] == 1 and o[10] == 9
    # stride-2 3x3 dgrad: 4 output-parity classes with 1+2+2+4 = 9 taps, all H*W pixels covered once
    assert lib.tdn_conv2d_plan(1, 2, 25, 42, 512, 512, 3, 2, 1, o) == 0
    assert o[9] == 4 and o[12] == 9 and o[0] == 2 * 25 * 42
    # 1x1 stride-2 dgrad: only the even/even class has a tap
    assert lib.tdn_conv2d_plan(1, 1, 50, 84, 512, 1024, 1, 2, 0, o) == 0
    assert o[9] == 4 and o[12] == 1 and o[10] == 1
    # wgrad: split-K over pixels
    assert lib.tdn_conv2d_plan(2, 2, 200, 336, 256, 256, 3, 1, 1, o) == 0
    assert (o[0], o[1], o[2]) == (256, 2304, 134400) and o[11] >= 8 and o[12] % 64 == 0
    assert o[11] * o[12] >= 134400
    # errors are reported, not thrown
    assert lib.tdn_conv2d_plan(0, 2, 25, 42, 500, 512, 3, 2, 1, o) != 0
    assert b"multiples of 64" in lib.tdn_last_error()
    assert lib.tdn_conv2d_plan(0, 1, 8, 8, 64, 64, 5, 1, 2, o) != 0


def test_launch_plan_bookkeeping_without_a_gpu():
    """tdn_plan_*: recording state machine and error reporting (no launches, no events: nothing touches a device)."""
    from torch_detection_amd import _lib
    lib = _lib.load()
    assert lib.tdn_plan_event_record(None) == -1            # nothing is being recorded
    assert lib.tdn_plan_begin() == 0
    assert lib.tdn_plan_begin() != 0 and b"already" in lib.tdn_last_error()
    plan = lib.tdn_plan_end()
    assert plan
    out = (ctypes.c_int32 * 3)()
    assert lib.tdn_plan_stats(plan, out) == 0 and list(out) == [0, 0, 0]
    assert lib.tdn_plan_free(plan) == 0
    assert not lib.tdn_plan_end() and b"no plan" in lib.tdn_last_error()
    assert lib.tdn_plan_run(None) != 0
