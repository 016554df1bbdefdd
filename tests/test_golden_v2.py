"""Golden vectors, second set (tests/golden/golden_v2.npz, made by tests/golden/make_golden_v2.py from the oracle): the oracle
must keep reproducing them (CPU) and the HIP path must match them bit for bit (GPU)."""
import os

import numpy as np
import pytest

import golden_v2_cases as cases

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v2.npz"))


def _check(g):
    assert set(g.keys()) == set(G.files)
    for k in G.files:
        a = np.ascontiguousarray(g[k]); b = np.ascontiguousarray(G[k])
        assert a.shape == b.shape and a.dtype == b.dtype, k
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), k


def test_oracle_reproduces_golden_v2(oracle):
    _check(cases.compute(cases.oracle_api(oracle), oracle))


@pytest.mark.gpu
def test_gpu_matches_golden_v2(oracle):
    from eorb_slam_amd import frontend as fe
    ctx = fe.Context()
    E = fe.EvImConverter

    class Api:
        @staticmethod
        def parse_events_text(text):
            return E.parse_events_text(text, ctx=ctx)

        @staticmethod
        def undistort_events(raw, mx, my, W, H, check, tsf):
            E.set_undistort_maps(mx, my, check, ctx=ctx)
            return E.undistort_events(raw, W, H, tsf, ctx=ctx)

        @staticmethod
        def ev2im_gauss_raw(raw, mx, my, W, H, sigma, pol, normalized):
            E.set_undistort_maps(mx, my, True, ctx=ctx)
            return E.ev2im_gauss_raw(raw, W, H, sigma, pol, normalized, ctx=ctx, return_all=True)

        @staticmethod
        def bow_transform(voc, desc, levelsup, weighting, norm):
            return fe.ORBVocabulary(voc, weighting, norm, ctx=ctx).transform(desc, levelsup, return_assignments=True)

        @staticmethod
        def search_by_bow_kf(*a):
            return fe.SearchByBoW_KF(*a, ctx=ctx)

        @staticmethod
        def search_for_triangulation(k1, d1, e1, fv1, k2, d2, e2, fv2, ep, F, scale, sig2, coarse, ori):
            n, pairs = fe.SearchForTriangulation(k1, d1, e1, fv1, k2, d2, e2, fv2, ep, F, scale, sig2, coarse, ori, ctx=ctx)
            m = np.full(len(k1), -1, np.int32); m[pairs[:, 0]] = pairs[:, 1]
            return n, m

        @staticmethod
        def kf_radius_match(kps, desc, valid, uv, radius, level, qd, inv_sigma2):
            return fe.KeyFrameRadiusMatch(kps, desc, fe.grid_bounds(cases.W, cases.H), valid, uv, radius, level, qd, inv_sigma2=inv_sigma2, ctx=ctx)

        @staticmethod
        def distinctive_descriptors(d, offs):
            return fe.ComputeDistinctiveDescriptors(d, offs, ctx=ctx)

        @staticmethod
        def calc_optical_flow_pyr_lk(i1, i2, pts, guess, win, lv, it, eps, flags):
            return fe.ELK_Tracker(win, lv, it, eps, ctx=ctx).calcOpticalFlowPyrLK(i1, i2, pts, guess, flags)
    _check(cases.compute(Api, oracle))
    ctx.close()
