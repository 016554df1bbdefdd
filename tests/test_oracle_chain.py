"""CPU tests of the oracle's L1 chain (oracle/orc_chain.py): known answers of the decision rules of EvImBuilder::Track /
generateMCImage, derived by hand from the reference source, and the shape of a whole sequence."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import chain_cases as cc


def test_contest_known_answers(oracle):
    from oracle import orc_chain
    W, H = 240, 180
    l2 = oracle.OrbExtractor(800, 1.0, 1, 0, 0, edgeTh=9, imWidth=W)
    evs = cc.stream(n_chunks=1, chunk=6001, seed=3, motion=6.0)
    p = cc.mci_poses(evs)
    # no pose available: only "EH" takes part, wins, and is rebuilt from the later half: evs.end() - evs.size() / 2 (:1062-1067, :1214-1216)
    r = orc_chain.generate_mc_image(evs, W, H, 1.0, cc.CAM, None, l2)
    assert r["winner"] == 2 and list(r["focus"][[0, 1, 3]]) == [-1, -1, -1]
    f32, _, _ = oracle.ev2im_gauss(evs[len(evs) - 3000:], W, H, 1.0, False, False)
    assert np.array_equal(r["image"], oracle.cv_normalize_minmax_u8(f32))
    assert r["focus"][4] == np.float32(oracle.measure_image_focus(f32))
    f32w, _, _ = oracle.ev2im_gauss(evs, W, H, 1.0, False, False)
    assert r["focus"][2] == np.float32(oracle.measure_image_focus(f32w))
    # equal keys: std::multimap keeps insertion order and begin() is the first of them -> "DP" before "BA" (:1207-1213)
    r = orc_chain.generate_mc_image(evs, W, H, 1.0, cc.CAM, dict(dp=p["dp"], ba=p["dp"]), l2)
    assert r["focus"][0] == r["focus"][1]
    assert r["winner"] in (0, 2) and (r["winner"] == 0) == bool(r["focus"][0] > r["focus"][2])
    # the winner is the largest focus
    r = orc_chain.generate_mc_image(evs, W, H, 1.0, cc.CAM, p, l2)
    assert r["winner"] == int(np.argmax(r["focus"][:4]))
    assert len(r["l2_kps"]) > 0


def test_window_size_rule_known_answers(oracle):
    """resolveEvWinSize / calcNewL1ChunkSize (:197-232): medPxDisp > maxPixelDisp on the idx-th tracked chunk gives
    floor(((idx + 1) / medPxDisp) * l1ChunkSize) events per chunk from then on, in float."""
    from oracle import orc_chain
    o = orc_chain.L1Chain(240, 180, cam=cc.CAM)
    res = orc_chain.run_sequence(o.track, cc.stream(n_chunks=30, chunk=2000, seed=5), 2000, cc.mci_poses)
    assert res[0]["state"] == orc_chain.INIT and "kps" in res[0] and len(res[0]["kps"]) > 100
    assert res[1]["state"] == orc_chain.TRACKING
    disp = [i for i, r in enumerate(res) if r["dispatched"]]
    assert disp, "the sequence never dispatched"
    i = disp[0]
    idx_after = i + 1                                            # chunks 0 .. i all went through updateState
    want = int(np.floor(np.float32(np.float32(idx_after + 1) / np.float32(res[i]["medPxDisp"])) * np.float32(2000)))
    assert res[i]["medPxDisp"] > 3.0 and res[i]["chunk_size"] == want
    assert res[i]["window"] == 2000 * (i + 1) and len(res[i]["overlap"]) == res[i]["window"] // 2
    assert res[i + 1]["state"] == orc_chain.INIT                # the next chunk re-initialises on the overlap's events
    assert np.array_equal(res[i]["overlap"]["ts"], np.concatenate([cc.stream(n_chunks=30, chunk=2000, seed=5)])[res[i]["window"] // 2:res[i]["window"]]["ts"])
