"""Child process of tests/test_gpu_parity.py::test_two_ranks_on_one_gpu_gather_their_records: one rank of a gloo group started by
shard.spawn_ranks; every rank runs ITS sequence's batch through the HIP front end on the (shared) GPU and the packed keypoint records
are gathered to rank 0, exactly as bench.py --gpus N does (RCCL itself refuses two ranks on one device: the collective's transport is
the only part of the multi-GPU path this rehearsal does not cover)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eorb_slam_amd import frontend as fe, shard, synth  # noqa: E402

W, H, B, N = 240, 180, 3, 30000


def run_rank(rank, dev):
    """rank's sequence through its own context; returns the packed record buffer (device) and the layout"""
    mx, my = synth.undistort_lut(W, H)
    raws = [synth.shapes_events(N, W, H, seed=900 + 10 * rank + b, motion=0.4, undistort=True, return_raw=True)[1] for b in range(B)]
    stream = torch.cuda.Stream(device=dev)
    ctx = fe.Context(device=0, stream=stream.cuda_stream)
    fb = fe.FrontEndBatch(W, H, 1.0, False, 1000, 1.2, 4, 10, 0, 19, max_batch=B, max_events=N, ctx=ctx)
    fe.EvImConverter.set_undistort_maps(mx, my, True, ctx=ctx)
    lay = shard.RecordLayout(B, fb.cap)
    rec = lay.alloc(dev)
    n_v, kp_v, desc_v = lay.views(rec)
    d_ev = torch.from_numpy(np.concatenate(raws).view(np.uint8)).to(dev)
    with torch.cuda.stream(stream):
        fb.run_dev(d_ev.data_ptr(), np.arange(B + 1, dtype=np.int64) * N, None, kp_v.data_ptr(), desc_v.data_ptr(), n_v.data_ptr(), raw=True)
    ctx.sync()
    return rec, lay, stream, ctx


def main():
    out_path = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=240))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    rec, lay, stream, ctx = run_rank(rank, dev)
    with torch.cuda.stream(stream):
        got = shard.gather_packed(rec, 0, None, "gloo")
    ok = True
    if rank == 0:
        ok = got is not None and len(got) == world
        for r in range(world):
            # what rank r's sequence gives in THIS process, alone
            ref, _, _, c2 = run_rank(r, dev)
            n, kp, desc = lay.unpack(got[r], synth.KP_DTYPE)
            rn, rkp, rdesc = lay.unpack(ref, synth.KP_DTYPE)
            ok = ok and np.array_equal(n, rn) and int(n.min()) > 20
            for b in range(B):                       # (rows past a slice's count are unwritten capacity)
                ok = ok and np.array_equal(kp[b, :n[b]].view(np.uint8), rkp[b, :n[b]].view(np.uint8)) and np.array_equal(desc[b, :n[b]], rdesc[b, :n[b]])
            c2.close()
        with open(out_path, "w") as f:
            f.write("ok %d" % world if ok else "bad")
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
