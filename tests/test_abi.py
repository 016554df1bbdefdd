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


def test_slot_gather_has_no_static_lds_and_no_scratch():
    """sl_gather_kernel addresses its LDS rows by byte offsets built with v_perm_b32 ({slot, 4 * lane}): the row table must start at
    LDS offset 0, i.e. the kernel may own no static LDS object (the dynamic segment then starts at 0; the kernel also checks this at run
    time and raises a flag), and it must not spill."""
    text = _device_asm("ev_slots.hip")
    blocks = re.findall(r"\.amdhsa_kernel (\S*sl_gather_kernel\S*)(.*?)\.end_amdhsa_kernel", text, re.S)
    assert len(blocks) == 1
    body = blocks[0][1]
    assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", body) and re.search(r"\.amdhsa_group_segment_fixed_size 0\b", body), body[:600]
    vg = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
    assert vg <= 64, "sl_gather_kernel needs %d VGPRs: fewer than 8 waves per SIMD" % vg


def test_hot_kernel_jump_arithmetic(tmp_path):
    """sl_hot_kernel enters its 64-entry sequence in the middle for the tail of a list: s_getpc + 20 bytes + 16 bytes per entry dword
    (tools/gen_sl_hot.py).  Both constants are encoding sizes: checked here against the assembled code object."""
    llvm = "/opt/rocm/lib/llvm/bin"
    if not (os.path.exists("/opt/rocm/bin/hipcc") and os.path.exists(llvm + "/llvm-objdump") and os.path.exists(llvm + "/clang-offload-bundler")):
        pytest.skip("no ROCm LLVM tools")
    src = os.path.join(ROOT, "eorb_slam_amd", "csrc", "ev_slots.hip")
    obj, co = str(tmp_path / "slots.o"), str(tmp_path / "slots.co")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-w", "-c", src, "-o", obj])
    subprocess.check_call([llvm + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + obj, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    dis = subprocess.run([llvm + "/llvm-objdump", "-d", "--mcpu=gfx950", co], capture_output=True, text=True).stdout.split("\n")
    ins = [(int(m.group(2), 16), m.group(1).strip()) for m in (re.match(r"\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", l) for l in dis) if m]
    gp = [i for i, (a, t) in enumerate(ins) if t.startswith("s_getpc_b64 s[34:35]")]
    assert len(gp) == 1
    i = gp[0]
    sp = next(j for j in range(i, i + 12) if ins[j][1].startswith("s_setpc_b64"))
    first = ins[sp + 1]
    assert first[1].startswith("s_mov_b32 m0, s36") and first[0] - ins[i + 1][0] == 20, (ins[i + 1], first)
    starts = [a for a, t in ins[sp + 1:sp + 1 + 32 * 4] if re.fullmatch(r"s_mov_b32 m0, s(3[6-9]|[45][0-9]|6[0-7])", t)]
    assert len(starts) == 32 and all(b - a == 16 for a, b in zip(starts, starts[1:])), starts
    seq = [t for _, t in ins[sp + 1:sp + 1 + 32 * 4]]
    assert all(seq[4 * d + 1].startswith("v_add_f32") and seq[4 * d + 3].startswith("v_add_f32")
               and re.fullmatch(r"s_lshr_b32 m0, s%d, 16" % (36 + d), seq[4 * d + 2]) for d in range(32)), seq[:8]
    # the kernel owns the whole architectural VGPR file of a wave: 256 registers, no scratch
    text = _device_asm("ev_slots.hip")
    body = re.findall(r"\.amdhsa_kernel (\S*sl_hot_kernel\S*)(.*?)\.end_amdhsa_kernel", text, re.S)[0][1]
    assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", body) and int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1)) == 256


def _device_asm(src_name):
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "eorb_slam_amd", "csrc", src_name)
    flags = re.search(r"^FLAGS\s*=\s*(.*?)\n\n", open(os.path.join(ROOT, "eorb_slam_amd", "csrc", "Makefile")).read(), re.S | re.M).group(1)
    flags = [f for f in flags.replace("\\\n", " ").split() if f not in ("-shared", "-fPIC") and not f.startswith("--offload-arch")]
    p = subprocess.run([hipcc] + flags + ["--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", "-", src], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    return p.stdout


def _vregs(tok):
    """VGPR numbers named by one operand token (v7, v[4:7]); nothing for other operands."""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def _written_vregs(line):
    """VGPRs an instruction writes (first operand of vector ALU ops and of LDS / memory ops that return data)."""
    parts = line.split(None, 1)
    if len(parts) < 2:
        return set()
    op, ops = parts[0], [o.strip() for o in parts[1].split(",")]
    if op.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_nop")):
        return set()
    if op.startswith("v_swap"):
        return _vregs(ops[0]) | _vregs(ops[1])
    if op.startswith("v_") or op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return _vregs(ops[0])
    if op.startswith("ds_") and (op.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle", "ds_consume", "ds_append")) or "_rtn" in op):
        return _vregs(ops[0])
    if op.startswith(("global_atomic", "buffer_atomic", "flat_atomic")) and " sc0" in line:
        return _vregs(ops[0])
    return set()


def test_raw_gather_asm_contract():
    """ev_gather_raw_kernel issues its table / entry loads from inline asm two batches ahead and waits for them with hand-counted
    s_waitcnt vmcnt(N).  That is only sound while (a) the loop carries exactly those loads -- 1 + 2 * NC per batch, two batches per
    unrolled iteration -- and no other vector-memory operation, (b) the compiler adds no vmcnt wait of its own inside the loop, (c) no
    instruction outside the asm statements writes a register while a load aimed at it is in flight, inside the loop or between the
    loop and the drain, and (d) that drain -- s_waitcnt vmcnt(0) in an asm statement -- follows the loop.  A late return into a
    reused register was a memory fault in round 2; this pins the generated code of all four instantiations."""
    text = _device_asm("ev_accum.hip")
    found = 0
    for pol in (0, 1):
        for nc in (2, 4):
            name = "_ZN4eorb20ev_gather_raw_kernelILb%dELi%dEEE" % (pol, nc)
            m = re.search(r"^(%s\S*):.*?\n(.*?)^\.Lfunc_end\d+:" % re.escape(name), text, re.S | re.M)
            assert m, name
            found += 1
            lines = []                                  # (text, in_asm)
            in_asm = False
            for ln in m.group(2).split("\n"):
                t = ln.split(";")[0].strip() if not ln.strip().startswith(";;#") else ln.strip()
                if t.startswith(";;#ASMSTART"):
                    in_asm = True; continue
                if t.startswith(";;#ASMEND"):
                    in_asm = False; continue
                if t:
                    lines.append((t, in_asm))
            label_at = {t[:-1]: i for i, (t, _) in enumerate(lines) if re.fullmatch(r"\.LBB\d+_\d+:", t)}
            is_load = [a and t.startswith("global_load") for t, a in lines]
            back = [(i, label_at[t.split()[-1]]) for i, (t, _) in enumerate(lines)
                    if t.startswith(("s_cbranch", "s_branch")) and t.split()[-1] in label_at and label_at[t.split()[-1]] < i
                    and any(is_load[label_at[t.split()[-1]]:i])           # back edges of THE loop: the ones around the asm loads ...
                    and not any(u.startswith("s_endpgm") for u, _ in lines[label_at[t.split()[-1]]:label_at[t.split()[-1]] + 3])]   # ... not jumps to the exit block
            assert back, name + ": no loop found"
            lo, hi = min(b[1] for b in back), max(b[0] for b in back)
            loop = lines[lo:hi + 1]
            loads = [t for t, a in loop if a and t.startswith("global_load")]
            assert len(loads) == 2 * (1 + 2 * nc), (name, len(loads))
            targets = set()
            for t in loads:
                targets |= _vregs(t.split(None, 1)[1].split(",")[0].strip())
            assert len(targets) == 2 * (2 + 2 * nc * 4), (name, sorted(targets))     # two register sets: an 8-byte entry + 2 * NC 16-byte pieces
            for t, a in loop:
                if a:
                    continue
                assert not t.startswith(("global_", "buffer_", "flat_", "scratch_")), (name, "vector-memory operation outside the asm statements", t)
                assert not (t.startswith("s_waitcnt") and "vmcnt" in t), (name, "compiler-inserted vmcnt wait inside the loop", t)
            hand = [t for t, a in loop if a and t.startswith("s_waitcnt") and "vmcnt" in t]
            assert sorted(set(hand)) == sorted({"s_waitcnt vmcnt(%d)" % (2 * nc + 1), "s_waitcnt vmcnt(%d)" % (2 * nc)}) and len(hand) == 4, (name, hand)
            # (c), (d): walk the code in layout order -- prologue, the loop twice (the second pass starts from the state the first one
            # leaves), the code behind the loop up to the drain -- with the queue of loads in flight (vmcnt retires in order).  Only
            # code that can run after the first load counts (the early exit of an empty tile is laid out in the middle).
            first = is_load.index(True)
            reach, work = set(), [first]
            while work:
                i = work.pop()
                while i < len(lines) and i not in reach:
                    reach.add(i)
                    t = lines[i][0]
                    if t.startswith(("s_cbranch", "s_branch")) and t.split()[-1] in label_at:
                        work.append(label_at[t.split()[-1]])
                    if t.startswith(("s_branch", "s_endpgm")):
                        break
                    i += 1
            inflight = []                               # target sets, oldest first
            drained = False
            order = list(range(first, hi + 1)) + list(range(lo, hi + 1)) + list(range(hi + 1, len(lines)))
            n_after = (hi + 1 - first) + (hi + 1 - lo)
            for k, i in enumerate(order):
                if i not in reach:
                    continue
                t, a = lines[i]
                if a and t.startswith("global_load"):
                    inflight.append(_vregs(t.split(None, 1)[1].split(",")[0].strip()))
                elif a and t.startswith("s_waitcnt") and "vmcnt" in t:
                    n = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
                    while len(inflight) > n:
                        inflight.pop(0)
                    if k >= n_after and n == 0:
                        drained = True; break
                elif not a:
                    busy = set().union(*inflight) if inflight else set()
                    assert not (_written_vregs(t) & busy), (name, "an instruction outside the asm statements writes the target of a load in flight", t)
            assert drained, name
    assert found == 4


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
