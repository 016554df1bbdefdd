"""Multi-GPU layout of the front end: independent sequences / contiguous time-slice ranges
are dealt one per rank (one process per GPU); there is no collective on the data path.  The only exchange
is the final gather of fixed-capacity per-slice keypoint records to rank 0 (RCCL over xGMI on GPUs, gloo in
the CPU tests and in single-GPU rehearsals).  bench.py and tests/test_host_logic.py both go through
RecordLayout / gather_packed below."""
import os
import socket
import subprocess
import sys

import numpy as np
import torch
import torch.distributed as dist

KP_BYTES, DESC_BYTES = 28, 32          # cv::KeyPoint, one ORB descriptor row


def slice_range(n_slices, rank, world, halo=1):
    """Contiguous range of time-slices for `rank` plus `halo` slices of overlap on the left (frame-to-frame
    matching needs the previous slice's descriptors).  Returns (first_owned, last_owned_exclusive, first_loaded)."""
    base, rem = divmod(n_slices, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi, max(lo - halo, 0)


def sequence_for_rank(n_sequences, rank, world):
    """C5: sequence s goes to rank s % world."""
    return [s for s in range(n_sequences) if s % world == rank]


class RecordLayout:
    """The keypoint records of one batch in ONE packed byte buffer {n[B] | kp[B][cap] | desc[B][cap][32]} (sections aligned to
    256 B), so that a step issues one collective however many arrays the records have."""

    def __init__(self, B, cap):
        self.B, self.cap = int(B), int(cap)
        self.off_n = 0
        self.off_kp = (self.B * 4 + 255) & ~255
        self.off_desc = (self.off_kp + self.B * self.cap * KP_BYTES + 255) & ~255
        self.nbytes = self.off_desc + self.B * self.cap * DESC_BYTES

    def alloc(self, device):
        return torch.zeros(self.nbytes, dtype=torch.uint8, device=device)

    def views(self, rec):
        """(n int32[B], kp uint8[B*cap*28], desc uint8[B*cap*32]) views into a packed torch buffer."""
        return (rec[:self.B * 4].view(torch.int32), rec[self.off_kp:self.off_kp + self.B * self.cap * KP_BYTES],
                rec[self.off_desc:self.off_desc + self.B * self.cap * DESC_BYTES])

    def unpack(self, rec, kp_dtype):
        """numpy (n[B], kps[B][cap] structured, desc[B][cap][32]) from a packed host buffer (torch CPU tensor or bytes-like)."""
        a = rec.cpu().numpy() if isinstance(rec, torch.Tensor) else np.frombuffer(rec, np.uint8)
        n = a[:self.B * 4].view(np.int32).copy()
        kp = a[self.off_kp:self.off_kp + self.B * self.cap * KP_BYTES].view(kp_dtype).reshape(self.B, self.cap).copy()
        desc = a[self.off_desc:self.off_desc + self.B * self.cap * DESC_BYTES].reshape(self.B, self.cap, DESC_BYTES).copy()
        return n, kp, desc


def gather_packed(rec, dst=0, recv=None, backend=None):
    """The path's one exchange: gather every rank's packed record buffer to `dst`.  On the nccl (= RCCL) backend the device
    buffer is sent as it is and the collective runs on torch's CURRENT stream (call it inside `with torch.cuda.stream(s)` of the
    stream that produced `rec`); on gloo the records are staged through host memory.  `recv`: optional preallocated list of
    world tensors on dst (reused across steps).  Returns the list on dst, None elsewhere (and [rec] without a process group)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [rec]
    world, rank = dist.get_world_size(), dist.get_rank()
    backend = backend or dist.get_backend()
    src = rec if backend == "nccl" else rec.cpu()
    if rank == dst and recv is None:
        recv = [torch.empty_like(src) for _ in range(world)]
    dist.gather(src, recv if rank == dst else None, dst=dst)
    return recv if rank == dst else None


def gather_records(tensors, dst=0):
    """Gather same-shaped per-rank tensors (keypoints, descriptors, counts) to `dst`, one collective per tensor.
    Returns a list (one entry per input tensor) of per-rank tensor lists on dst, None elsewhere."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    out = []
    for t in tensors:
        if world == 1:
            out.append([t])
            continue
        lst = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
        dist.gather(t, lst, dst=dst)
        out.append(lst)
    return out if rank == dst else None


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def spawn_ranks(script, argv, n, extra_env=None, timeout_s=None):
    """Start `n` ranks of `script` as CHILD processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) and return
    the first non-zero exit code (0 when all succeed).  All children are polled: when one dies (no GPU, out of memory, a failed
    RCCL init) its siblings -- which would otherwise wait for it in a barrier or a gather for ever -- are terminated, then killed;
    the same happens after `timeout_s` seconds (default: EORB_SPAWN_TIMEOUT or 1800).  The caller must not have touched the GPU:
    nothing is exec'ed over a process that initialised HIP, the children are ordinary subprocesses and each initialises its own
    device.  (The rendezvous port is picked by binding port 0 and released just before the children start; a child that finds it
    taken fails its init and brings the launch down with a non-zero code instead of hanging.)"""
    import time
    if timeout_s is None:
        timeout_s = float(os.environ.get("EORB_SPAWN_TIMEOUT", "1800"))
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env))
    t0, rc = time.time(), 0
    live = list(procs)
    while live and rc == 0:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = abs(code) or 1
                break
        if rc == 0 and live:
            if time.time() - t0 > timeout_s:
                print("spawn_ranks: %d rank(s) still running after %.0f s, stopping them" % (len(live), timeout_s), file=sys.stderr)
                rc = 124
                break
            time.sleep(0.05)
    if rc != 0:
        for p in live:
            p.terminate()
        t1 = time.time()
        for p in live:
            try:
                p.wait(timeout=max(0.1, 10 - (time.time() - t1)))
            except subprocess.TimeoutExpired:
                p.kill(); p.wait()
    return rc
