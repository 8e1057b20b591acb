"""One rank of a PeerExchange self-test (kinovagrasping_amd/exchange.py): RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the
environment (torch.distributed env:// rendezvous), backend KS_DIST_BACKEND (default gloo: lets several ranks share one GPU).
tests/test_bench_launch.py starts 2 and 4 of these on the box's GPU; on a multi-GPU node run it under torch.distributed.run
with KS_DIST_BACKEND=nccl to check peer access over xGMI."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from kinovagrasping_amd.exchange import try_peer_exchange

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
ngpu = torch.cuda.device_count()
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", rank)) % ngpu)
torch.cuda.set_device(dev)
backend = os.environ.get("KS_DIST_BACKEND", "gloo")
if backend == "nccl":
    dist.init_process_group("nccl", device_id=dev)
else:
    dist.init_process_group(backend)
counts = (88068, 88321)                      # the 256-256 actor's and critic's flat gradient buffers
ex, why = try_peer_exchange(max(counts))
if ex is None:
    print(f"rank {rank}: NO PEER EXCHANGE ({why})", flush=True)
    sys.exit(3)
# the learner's pattern: critic, actor, critic, ... on a side stream, many epochs, results against the exact fp64 mean
side = torch.cuda.Stream(dev)
worst = 0.0
for it in range(int(os.environ.get("KS_XCHG_ROUNDS", "10"))):
    for n in counts[::-1]:
        g = torch.Generator(device="cpu").manual_seed(1000 * it + n)
        every = torch.randn(world, n, generator=g)                  # every rank builds all ranks' inputs: the expectation is local
        mine = every[rank].to(dev)
        with torch.cuda.stream(side):
            ex.allreduce_mean(mine)
        side.synchronize()
        want = every[0].clone()
        for r in range(1, world):
            want += every[r]                                        # rank order, fp32: what the kernel does
        want *= 1.0 / world
        worst = max(worst, float((mine.cpu() - want).abs().max()))
        assert torch.equal(mine.cpu(), want), (it, n, worst)
assert ex.failed_epoch() == 0
dist.barrier()
ex.close()
print(f"rank {rank}/{world}: peer exchange OK (all-reduces bitwise equal to the rank-order fp32 mean)", flush=True)
dist.destroy_process_group()
