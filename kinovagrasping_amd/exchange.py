"""The learner's gradient exchange between the env-shard ranks without a library collective (csrc/ks_xchg.hip).

RCCL's all-reduce kernels need LDS; ks_step's stepping kernel holds all of every CU's LDS for the whole env-step, so a
collective issued beside it waits until stepping workgroups retire and the learner's chain ends up BEHIND the simulator.
`PeerExchange` is an LDS-free all-reduce over peer-mapped device memory (hipIpc handles gathered through the process group,
one 32-workgroup kernel per call that reads the peers' buffers directly - xGMI peer access on a node): it runs in the
stepping kernel's shadow like the rest of the update.  The sum runs in rank order on every rank, so replicas stay
bit-identical.  `connect()` ends with a self-test against the process group's own all_reduce; every rank takes
the same decision, and a rank set without working peer access keeps the library collective.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import sim as _sim

HANDLE_BYTES = 64


def _injected(stage: str, rank: int) -> bool:
    """KS_XCHG_INJECT="connect:<rank>" / "selftest:<rank>": make that rank fail that stage (tests/test_bench_launch.py proves that
    EVERY rank then falls back to the process group's all_reduce and says so - the first run on a real 8-GPU node must not hang
    or train diverged replicas whatever peer access turns out to do there)."""
    import os
    return os.environ.get("KS_XCHG_INJECT", "") == f"{stage}:{rank}"


class PeerExchange:
    def __init__(self, max_count: int, group=None, device=None):
        import torch.distributed as dist
        self.group, self.dist = group, dist
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.lib = _sim.load_library()
        self.x = C.c_void_p()
        self.max_count = int(max_count)
        handle = (C.c_uint8 * HANDLE_BYTES)()
        with torch.cuda.device(self.device):
            rc = self.lib.kr_xchg_create(C.byref(self.x), self.world, self.rank, self.max_count, handle)
        mine = (rc, bytes(handle))
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine, group=group)
        if any(r != 0 for r, _ in everyone):
            self.close()
            raise RuntimeError(f"kr_xchg_create failed on ranks {[i for i, (r, _) in enumerate(everyone) if r != 0]}")
        with torch.cuda.device(self.device):
            rc = self.lib.kr_xchg_connect(self.x, b"".join(h for _, h in everyone))
        if _injected("connect", self.rank):
            rc = -3             # fault injection (tests): this rank pretends hipIpcOpenMemHandle failed
        rcs = [None] * self.world
        dist.all_gather_object(rcs, rc, group=group)
        if any(r != 0 for r in rcs):
            self.close()
            raise RuntimeError(f"kr_xchg_connect (hipIpcOpenMemHandle) failed on ranks {[i for i, r in enumerate(rcs) if r != 0]}")

    def allreduce_mean(self, t: torch.Tensor):
        """t <- mean over the ranks, in place, asynchronous on the current stream; the same call on every rank"""
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() <= self.max_count
        rc = self.lib.kr_xchg_allreduce_mean(self.x, _sim._ptr(t), t.numel(), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"kr_xchg_allreduce_mean failed ({rc})")

    def failed_epoch(self) -> int:
        e = C.c_uint32(0)
        if self.lib.kr_xchg_status(self.x, C.byref(e)) != 0:
            raise RuntimeError("kr_xchg_status failed")
        return int(e.value)

    def check(self):
        """Raise if one of this rank's exchange kernels gave up waiting for a peer (KS_XCHG_TIMEOUT_S, 60 s by default).  From that
        call on this rank's gradients are no longer reduced while its peers may carry on: the replicas have diverged and the run
        must stop - loudly, on the training path (GraphedTrainer polls this every few hundred updates and at every flush)."""
        e = self.failed_epoch()
        if e:
            raise RuntimeError(f"PeerExchange: rank {self.rank} timed out waiting for a peer in all-reduce call {e}; its gradients have not been "
                               "reduced since - replicas have diverged.  (KS_XCHG_TIMEOUT_S sets the bound, <= 0 waits for ever; KS_P2P=0 uses "
                               "the library collective.)")

    def self_test(self, rounds: int = 3) -> str | None:
        """None when the exchange reproduces the process group's all_reduce on every rank, else the reason (same on all ranks)"""
        dist, dev = self.dist, self.device
        reason = None

        def note(text):
            nonlocal reason
            reason = reason or text

        # every rank walks the same sequence of collectives whatever it finds on the way (a rank that left the loop early
        # would leave its peers inside an all_reduce): findings are only recorded, and compared at the end
        g = torch.Generator(device="cpu").manual_seed(1234 + self.rank)
        for k in range(rounds):
            n = max(1, min(self.max_count, self.max_count - 3 * k - (k % 2)))     # also counts that are not multiples of 4
            a = torch.randn(n, generator=g).to(dev)
            ref = a.clone()
            try:
                self.allreduce_mean(a)
            except Exception as e:      # noqa: BLE001 - any failure means: keep the library collective
                note(f"{type(e).__name__}: {e}")
            dist.all_reduce(ref, op=dist.ReduceOp.SUM, group=self.group)
            ref /= self.world
            torch.cuda.synchronize(dev)
            try:
                if self.failed_epoch():
                    note(f"a peer did not arrive (call {self.failed_epoch()})")
            except Exception as e:      # noqa: BLE001
                note(f"{type(e).__name__}: {e}")
            if _injected("selftest", self.rank):
                a[0] += 1.0     # fault injection (tests): this rank's exchange "returns" a wrong mean
            if not torch.allclose(a, ref, rtol=1e-5, atol=1e-6):
                note(f"mismatch against all_reduce: {float((a - ref).abs().max()):.3e}")
            # bitwise identical on all ranks
            lo, hi = a.clone(), a.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
            if not torch.equal(lo, hi):
                note("ranks disagree bitwise")
        reasons = [None] * self.world
        dist.all_gather_object(reasons, reason, group=self.group)
        bad = [f"rank {i}: {r}" for i, r in enumerate(reasons) if r]
        return "; ".join(bad) if bad else None

    def close(self):
        if getattr(self, "x", None) is not None and self.x.value:
            self.lib.kr_xchg_destroy(self.x)
            self.x = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001
            pass


def try_peer_exchange(max_count: int, group=None, device=None):
    """(PeerExchange, None) when peer access works and the self-test passes on every rank, else (None, reason)"""
    try:
        ex = PeerExchange(max_count, group, device)
    except Exception as e:      # noqa: BLE001
        return None, f"{type(e).__name__}: {e}"
    reason = ex.self_test()
    if reason is not None:
        ex.close()
        return None, "self-test: " + reason
    return ex, None
