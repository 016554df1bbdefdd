"""Host-side logic that needs no GPU: synthetic generators, slice sharding and the 2-rank gather (gloo)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from eorb_slam_amd import shard, synth


def test_generators_are_deterministic_and_in_image():
    a = synth.shapes_events(5000, seed=7, undistort=True); b = synth.shapes_events(5000, seed=7, undistort=True)
    assert a.tobytes() == b.tobytes()
    assert (a["x"] >= 0).all() and (a["x"] < 240).all() and (a["y"] >= 0).all() and (a["y"] < 180).all()
    assert (np.diff(a["ts"]) > 0).all()
    c = synth.shapes_events(5000, seed=7)
    assert np.array_equal(c["x"], np.floor(c["x"])) and not np.array_equal(a["x"], np.floor(a["x"]))
    lx, ly = synth.undistort_lut()
    assert lx.shape == (180, 240) and abs(lx[110, 132] - 132) < 0.5 and abs(ly[110, 132] - 110) < 0.5
    img = synth.texture_image(240, 180, seed=3)
    assert img.shape == (180, 240) and img.dtype == np.uint8 and img.std() > 20


def test_slice_ranges_partition_and_halo():
    for n, w in ((64, 8), (10, 4), (3, 8), (1, 1)):
        owned = []
        for r in range(w):
            lo, hi, first = shard.slice_range(n, r, w)
            owned += list(range(lo, hi))
            assert first == max(lo - 1, 0)
        assert owned == list(range(n))
    assert shard.sequence_for_rank(8, 3, 8) == [3] and shard.sequence_for_rank(8, 1, 2) == [1, 3, 5, 7]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cap = 16
    # rank r owns sequences r, r+world...: fabricate its fixed-capacity keypoint records
    n = torch.tensor([3 + rank, 5 + rank], dtype=torch.int32)
    kp = torch.full((2, cap, 7), float(rank), dtype=torch.float32)
    desc = torch.full((2, cap, 32), rank, dtype=torch.uint8)
    got = shard.gather_records([n, kp, desc], dst=0)
    if rank == 0:
        ok = got is not None and len(got) == 3 and all(len(g) == world for g in got)
        for r in range(world):
            ok = ok and got[0][r].tolist() == [3 + r, 5 + r] and float(got[1][r].mean()) == float(r) and int(got[2][r].max()) == r
        q.put(ok)
    else:
        q.put(got is None)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    assert all(res)


def test_record_layout_round_trip():
    lay = shard.RecordLayout(5, 1032)
    assert lay.off_kp % 256 == 0 and lay.off_desc % 256 == 0 and lay.nbytes == lay.off_desc + 5 * 1032 * 32
    rec = lay.alloc("cpu")
    n, kp, desc = lay.views(rec)
    assert n.numel() == 5 and kp.numel() == 5 * 1032 * 28 and desc.numel() == 5 * 1032 * 32
    n.copy_(torch.arange(5, dtype=torch.int32)); desc.fill_(9)
    gn, gk, gd = lay.unpack(rec, synth.KP_DTYPE)
    assert gn.tolist() == [0, 1, 2, 3, 4] and gk.shape == (5, 1032) and int(gd.sum()) == 9 * 5 * 1032 * 32


def test_spawned_ranks_gather_packed_records(tmp_path):
    """The launcher behind `python bench.py --gpus N` (shard.spawn_ranks: N child processes with RANK / WORLD_SIZE / MASTER_* set)
    and the collective bench.py issues per step (shard.gather_packed on the packed record buffer), world size 2 on gloo."""
    out = tmp_path / "rank0.txt"
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_rank_worker.py")
    rc = shard.spawn_ranks(script, [str(out)], 2)
    assert rc == 0
    assert out.read_text() == "ok 2"


def test_spawned_rank_failure_stops_the_launch():
    """A rank that dies must not leave `bench.py --gpus N` hanging on its siblings: the launcher polls all children, returns the
    failing rank's code and stops the others; a launch that exceeds its time limit is stopped the same way."""
    import time
    from eorb_slam_amd import shard
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_rank_fail_worker.py")
    t0 = time.time()
    assert shard.spawn_ranks(script, [], 2) == 3
    assert time.time() - t0 < 30
    t0 = time.time()
    assert shard.spawn_ranks(script, [], 1, timeout_s=1.0) == 124          # (rank 0 alone sleeps: the time limit ends it)
    assert time.time() - t0 < 30


def test_bench_refuses_mismatched_world(monkeypatch):
    """`--gpus N` must agree with WORLD_SIZE when a launcher set it (the driver's torch.distributed.run form)."""
    import subprocess, sys
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr
