"""Build-container check: the constants the product and the oracle carry still equal the reference's source text.

/root/reference exists only in the build container (never on the GPU box), so every test here is skipped when it is absent.
Reading the reference as text is study: nothing is imported, compiled or copied from it."""
import os
import re

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")


def _read(*p):
    with open(os.path.join(*p)) as f:
        return f.read()


def _ints_of_table(text, name):
    i = text.index(name)
    body = text[text.index("{", i) + 1:text.index("};", i)]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    body = re.sub(r"//[^\n]*", "", body)
    return [int(x) for x in re.findall(r"-?\d+", body)]


def _const(text, pattern):
    m = re.search(pattern, text)
    assert m, pattern
    return float(m.group(1)) if "." in m.group(1) else int(m.group(1))


def test_brief_pattern_equals_reference_table():
    ref = _ints_of_table(_read(REF, "src/ORBextractor.cc"), "static int bit_pattern_31_[256*4]")      # :160-418
    assert len(ref) == 1024 and max(abs(v) for v in ref) == 13
    prod = _ints_of_table(_read(ROOT, "eorb_slam_amd/csrc/orb_pattern.h"), "k_orb_pattern_31[1024]")
    orc = _ints_of_table(_read(ROOT, "oracle/orc_pattern.h"), "orc_bit_pattern_31[1024]")
    assert prod == ref and orc == ref


def test_extractor_constants_equal_reference():
    ref = _read(REF, "src/ORBextractor.cc")                                                            # :71-74
    patch = _const(ref, r"const int PATCH_SIZE = (\d+);")
    half = _const(ref, r"const int HALF_PATCH_SIZE = (\d+);")
    edge = _const(ref, r"int EDGE_THRESHOLD = (\d+);")
    wden = _const(ref, r"const float W_denom = (\d+);")
    assert (patch, half, edge, wden) == (31, 15, 19, 30)
    orc = _read(ROOT, "oracle/orc_orb.c")
    assert _const(orc, r"#define PATCH_SIZE (\d+)") == patch and _const(orc, r"#define HALF_PATCH_SIZE (\d+)") == half
    # the adaptive edge rule (:481-488) and the cell width (:805-808) as literals in both implementations
    hdr = _read(REF, "include/ORBextractor.h")
    assert _const(hdr, r"#define DEF_IMAGE_WIDTH (\d+)") == 752
    prod = _read(ROOT, "eorb_slam_amd/csrc/orb_extract.hip")
    for src in (prod, orc):
        assert re.search(r"19 \* \(\(float\)\s*\w[\w>\-\.]*imWidth / \(float\)752\)", src), "adaptive edge rule 19*(imWidth/752)"
        assert re.search(r"31\.0f \* |\(float\)PATCH_SIZE \* ", src), "scaledPatchSize"
    assert re.search(r"width / 30\.0f", prod) and re.search(r"height / 30\.0f", prod), "W_denom (product)"
    assert _const(orc, r"#define W_DENOM (\d+)\.0f") == wden and "width / W_DENOM" in orc and "height / W_DENOM" in orc
    # HALF_PATCH_SIZE in IC_Angle (:77-104): columns -15 .. 15 (one lane each), row pairs 1 .. 15 inside the disc
    assert "au <= G->umax[v]" in prod and "(int)(threadIdx.x & 31) - 15" in prod and "for (int v = 1; v <= 15; ++v)" in prod and "if (au <= 15)" in prod


def test_matcher_constants_equal_reference():
    ref = _read(REF, "src/ORBmatcher.cc")                                                              # :36-38
    th = (_const(ref, r"ORBmatcher::TH_HIGH = (\d+);"), _const(ref, r"ORBmatcher::TH_LOW = (\d+);"),
          _const(ref, r"ORBmatcher::HISTO_LENGTH = (\d+);"))
    assert th == (100, 50, 30)
    prod = _read(ROOT, "eorb_slam_amd/csrc/match.hip")
    m = re.search(r"constexpr int TH_HIGH = (\d+), TH_LOW = (\d+), HISTO_LENGTH = (\d+);", prod)
    assert m and tuple(int(g) for g in m.groups()) == th
    orc = _read(ROOT, "oracle/orc_match.c")
    assert (_const(orc, r"#define TH_HIGH (\d+)"), _const(orc, r"#define TH_LOW (\d+)"), _const(orc, r"#define HISTO_LENGTH (\d+)")) == th
    mixed = _read(REF, "src/MixedMatcher.cpp")
    assert "TH_HIGH" in mixed and "HISTO_LENGTH" in mixed               # the Mixed variants use the same class constants


def test_frame_grid_equals_reference():
    ref = _read(REF, "include/Frame.h")                                                                # :45-46
    rows, cols = _const(ref, r"#define FRAME_GRID_ROWS (\d+)"), _const(ref, r"#define FRAME_GRID_COLS (\d+)")
    assert (rows, cols) == (48, 64)
    ctxh = _read(ROOT, "eorb_slam_amd/csrc/eorb_ctx.h")
    assert _const(ctxh, r"constexpr int kGridCols = (\d+);") == cols and _const(ctxh, r"constexpr int kGridRows = (\d+);") == rows
    orc = _read(ROOT, "oracle/orc_match.c")
    assert _const(orc, r"#define FRAME_GRID_ROWS (\d+)") == rows and _const(orc, r"#define FRAME_GRID_COLS (\d+)") == cols


def test_event_record_and_stamp_constants_equal_reference():
    # EventData = {double ts; float x, y; bool p} (include/Event/EventData.h:36-58): 24 bytes with padding
    ev = _read(REF, "include/Event/EventData.h")
    m = re.search(r"double ts = 0\.0;\s*float x = 0\.f;\s*float y = 0\.f;\s*bool p = false;", ev)        # member order = layout
    assert m, "EventData member order changed"
    api = _read(ROOT, "include/eorb_fe.h")
    assert re.search(r"double\s+ts;", api) and "eorb_event" in api
    # ev2im_gauss: half window = ceil(sigma * 3) and the 0.001f count increment (src/Event/EventConversion.cc:173-269)
    conv = _read(REF, "src/Event/EventConversion.cc")
    assert re.search(r"ceil\(sigma\s*\*\s*3(\.0)?\)", conv)
    assert "0.001" in conv
    orc = _read(ROOT, "oracle/orc_events.c")
    assert re.search(r"ceil\(\(double\)sigma \* 3\.0\)", orc) and "0.001f" in orc
    prod = _read(ROOT, "eorb_slam_amd/csrc/ev_accum.hip")
    assert re.search(r"ceil\(\(double\)sigma \* 3\.0\)", prod) and "0.001f" in prod
