"""Child process of tests/test_host_logic.py::test_spawned_ranks_gather_packed_records: one rank of a gloo group started by
shard.spawn_ranks (the launcher bench.py uses for `--gpus N` from a bare shell).  No GPU."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eorb_slam_amd import shard, synth  # noqa: E402


def main():
    out_path = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert int(os.environ["LOCAL_RANK"]) == rank and os.environ["MASTER_ADDR"] == "127.0.0.1"
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    B, cap = 3, 16
    lay = shard.RecordLayout(B, cap)
    rec = lay.alloc("cpu")
    n, kp, desc = lay.views(rec)
    # rank r owns sequence r: fabricate its fixed-capacity keypoint records
    n.copy_(torch.tensor([3 + rank, 5 + rank, 0], dtype=torch.int32))
    kps = np.zeros((B, cap), synth.KP_DTYPE); kps["x"] = rank + 0.5; kps["octave"] = rank
    kp.copy_(torch.from_numpy(kps.view(np.uint8).reshape(-1)))
    desc.fill_(17 * (rank + 1))
    got = shard.gather_packed(rec, dst=0)
    ok = True
    if rank == 0:
        ok = got is not None and len(got) == world
        for r in range(world):
            gn, gk, gd = lay.unpack(got[r], synth.KP_DTYPE)
            ok = ok and gn.tolist() == [3 + r, 5 + r, 0] and bool((gk["x"] == r + 0.5).all()) and bool((gk["octave"] == r).all())
            ok = ok and gd.shape == (B, cap, 32) and int(gd.min()) == int(gd.max()) == 17 * (r + 1)
        with open(out_path, "w") as f:
            f.write("ok %d" % world if ok else "bad")
    else:
        ok = got is None
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
