"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every symbol
that include/eorb_fe.h declares, and refuses to run without a GPU instead of falling back to the CPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "eorb_fe.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(eorb_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_every_declared_symbol():
    from eorb_slam_amd import _lib
    path = _lib.build()
    assert os.path.exists(path)
    L = C.CDLL(path)
    decl = _declared()
    assert len(decl) >= 20
    for sym in decl:
        assert hasattr(L, sym), "libeorb_fe.so does not export %s" % sym
    assert sorted(_lib.EXPORTS) == decl, "eorb_slam_amd/_lib.py EXPORTS out of sync with include/eorb_fe.h"
    L.eorb_version.restype = C.c_char_p
    assert b"gfx950" in L.eorb_version()


def test_library_contains_gfx950_code_object():
    from eorb_slam_amd import _lib
    data = open(_lib.build(), "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in data, "no gfx950 code object embedded in libeorb_fe.so"
    for kern in (b"ev_gather_kernel", b"ev_gather_raw_kernel", b"ev_count_kernel", b"ev_scan_kernel", b"ev_scatter_kernel", b"fast_cells_kernel", b"octree_kernel", b"brief_kernel",
                 b"win_cand_kernel", b"win_resolve_kernel", b"bf_knn2_kernel"):
        assert kern in data, kern


def test_raw_gather_kernels_use_no_scratch():
    """ev_gather_raw_kernel places its own s_waitcnt vmcnt(N) around loads issued from inline asm; a register spill (scratch
    traffic counts in vmcnt too) would silently break that count, so every instantiation must have a zero private segment."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "eorb_slam_amd", "csrc", "ev_accum.hip")
    flags = re.search(r"^FLAGS\s*=\s*(.*?)\n\n", open(os.path.join(ROOT, "eorb_slam_amd", "csrc", "Makefile")).read(), re.S | re.M).group(1)
    flags = [f for f in flags.replace("\\\n", " ").split() if f not in ("-shared", "-fPIC") and not f.startswith("--offload-arch")]
    p = subprocess.run([hipcc] + flags + ["--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", "-", src], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    blocks = re.findall(r"\.amdhsa_kernel (\S*ev_gather_raw_kernel\S*)(.*?)\.end_amdhsa_kernel", p.stdout, re.S)
    assert len(blocks) == 4, [b[0] for b in blocks]
    for name, body in blocks:
        assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", body), name + " spills to scratch"
        assert int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", body).group(1)) == 32768, name


def test_header_is_plain_c():
    """The boundary must be consumable from C (cgo/JNI/ctypes style bindings): compile it with gcc -std=c99."""
    src = '#include "eorb_fe.h"\nint main(void){ eorb_event e; eorb_keypoint k; (void)e; (void)k; return sizeof(eorb_event)==24 && sizeof(eorb_keypoint)==28 && sizeof(eorb_event16)==16 ? 0 : 1; }\n'
    exe = os.path.join(ROOT, "tests", "_abi_c_check")
    p = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-x", "c", "-", "-o", exe],
                       input=src, text=True, capture_output=True)
    assert p.returncode == 0, p.stderr
    assert subprocess.run([exe]).returncode == 0
    os.remove(exe)


def test_no_cpu_fallback_without_gpu():
    """Without a HIP device the product path must fail loudly (never route through the oracle)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from eorb_slam_amd import frontend
    with pytest.raises(frontend.EorbError):
        frontend.Context()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "eorb_slam_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_py" not in txt and "liboracle" not in txt and "eorb_oracle.h" not in txt, f


def test_pack_events_layout():
    from eorb_slam_amd import frontend, synth
    ev = synth.random_events(100, seed=1)
    p = frontend.pack_events(ev)
    assert p.dtype.itemsize == 16 and np.array_equal(p["x"], ev["x"]) and np.array_equal(p["y"], ev["y"])
    neg = np.signbit(p["t"])
    assert np.array_equal(neg, ev["p"] == 0) and np.array_equal(np.abs(p["t"]), ev["ts"])
