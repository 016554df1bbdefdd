"""Hygiene of the test infrastructure itself (no GPU): the two builds of the oracle agree, its C code is clean under
AddressSanitizer / UndefinedBehaviorSanitizer on the known-answer and golden tests, and the C++ host mirror's context pool is clean
under ThreadSanitizer ("Race detection / sanitizers": the reference admits data races, README.md:95)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_timing_build_equals_parity_build():
    """liboracle_fast.so (-O3 -march=native, bench.py's cpu_baseline and the full-size GPU tests) against liboracle.so (-O2, the
    parity build): one 200 000-event slice -> f32 image, extremes, u8 image; ORB-1000 on two such images; SearchForInitialization
    between them; brute-force 2-NN.  Bit for bit."""
    from oracle import oracle_py as o
    from eorb_slam_amd import synth
    W, H = 240, 180
    frames = []
    for seed in (3, 4):
        ev = synth.shapes_events(200000, W, H, seed=seed, undistort=True)
        a = o.ev2im_gauss(ev, W, H, 1.0, False, True, fast=False)
        b = o.ev2im_gauss(ev, W, H, 1.0, False, True, fast=True)
        assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1], b[1])
        assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
        ap = o.ev2im_gauss(ev[:50000], W, H, 0.7, True, True, fast=False)
        bp = o.ev2im_gauss(ev[:50000], W, H, 0.7, True, True, fast=True)
        assert np.array_equal(ap[0].view(np.uint32), bp[0].view(np.uint32)) and np.array_equal(ap[2].view(np.uint32), bp[2].view(np.uint32))
        res = []
        for fast in (False, True):
            oe = o.OrbExtractor(1000, 1.2, 4, 10, 0, edgeTh=19, fast=fast)
            res.append(oe.extract(a[1]))
        assert res[0][0] == res[1][0] and np.array_equal(res[0][1].view(np.uint8), res[1][1].view(np.uint8))
        assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])
        assert len(res[0][1]) > 50
        frames.append((res[0][1], res[0][2]))
    out = []
    for fast in (False, True):
        F1 = o.Frame(frames[0][0], frames[0][1], W, H, fast=fast); F2 = o.Frame(frames[1][0], frames[1][1], W, H, fast=fast)
        pm = np.stack([frames[0][0]["x"], frames[0][0]["y"]], axis=1)
        n, m12, pmo = o.search_for_initialization(F1, F2, pm, 100, 0.9, True)
        q = synth.random_descriptors(300, seed=1); t = synth.random_descriptors(500, seed=2)
        out.append((n, m12.copy(), o.bf_knn2(q, t, fast=fast)))
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])
    for x, y in zip(out[0][2], out[1][2]):
        assert np.array_equal(x, y)


def _sanitizer_lib(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_is_clean_under_asan_and_ubsan():
    """make -C oracle asan, then the known-answer and golden tests of the oracle run against that build (every ctypes call into it:
    heap / stack overflows, use after free, signed overflow, misaligned or out-of-range accesses abort the child process)."""
    libasan = _sanitizer_lib("libasan.so")
    if not libasan:
        pytest.skip("no libasan")
    env = dict(os.environ, EORB_ORACLE_VARIANT="asan", LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_kat.py"), os.path.join(ROOT, "tests", "test_golden.py"),
                        os.path.join(ROOT, "tests", "test_golden_v2.py")], capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0, tail
    assert "AddressSanitizer" not in p.stdout + p.stderr and "runtime error" not in p.stdout + p.stderr, tail
    assert " passed" in p.stdout


def test_host_context_pool_is_clean_under_tsan(tmp_path):
    """eorb_host::ContextPool (borrowed contexts of the reference's transient converter threads, maps replaced meanwhile, one injected
    upload failure) against a stub of the five C-ABI entry points it calls, built with -fsanitize=thread."""
    if not _sanitizer_lib("libtsan.so"):
        pytest.skip("no libtsan")
    exe = str(tmp_path / "pool_tsan")
    src = os.path.join(ROOT, "tests", "host")
    subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=thread", "-c", "-o", str(tmp_path / "stub.o"), os.path.join(src, "eorb_stub.c")])
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-o", exe, os.path.join(src, "pool_tsan.cpp"), str(tmp_path / "stub.o")])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert p.returncode == 0, (p.stdout, p.stderr[-3000:])
    assert "ThreadSanitizer" not in p.stderr and "failures=1 inconsistent=0" in p.stdout, (p.stdout, p.stderr[-2000:])
