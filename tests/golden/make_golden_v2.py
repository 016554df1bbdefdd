#!/usr/bin/env python3
"""Golden vectors, second set: the rows added after golden_v1 (loader text + rectification, raw-event images, DBoW2
transform, KeyFrame-side matchers, pyramidal LK).  Same provenance as golden_v1 (see make_golden.py): produced by the CPU
oracle from seeded synthetic inputs; tests/test_golden_v2.py checks the oracle (CPU) and the HIP path (GPU) against them.

Run from the repository root:  python tests/golden/make_golden_v2.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from eorb_slam_amd import synth  # noqa: E402
from oracle import oracle_py as orc  # noqa: E402
import golden_v2_cases as cases  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    g = cases.compute(cases.oracle_api(orc), orc)
    np.savez_compressed(os.path.join(OUT, "golden_v2.npz"), **g)
    print("wrote golden_v2.npz:", os.path.getsize(os.path.join(OUT, "golden_v2.npz")), "bytes,", len(g), "arrays")


if __name__ == "__main__":
    main()
