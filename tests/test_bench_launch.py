"""bench.py as a launcher: `--gpus N` starts N ranks itself (children spawned before any torch / HIP call), refuses a
WORLD_SIZE that contradicts --gpus, and the N > 1 line carries n_gpus, the RCCL block and the replica checksum spread."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def run_bench(args, env=None, timeout=900):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, env=e, capture_output=True, text=True, timeout=timeout)


def test_gpus_flag_must_match_world_size():
    """under a launcher (WORLD_SIZE set) a contradicting --gpus fails fast, before torch is imported"""
    r = run_bench(["--gpus", "1"], env={"WORLD_SIZE": "2", "RANK": "0"}, timeout=60)
    assert r.returncode != 0 and "--gpus 1 but WORLD_SIZE=2" in r.stderr
    r = run_bench(["--gpus", "4"], env={"WORLD_SIZE": "2", "RANK": "0"}, timeout=60)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_launcher_spawns_ranks_before_touching_torch(tmp_path):
    """launch_ranks: N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* on 127.0.0.1; the parent never imports torch"""
    probe = tmp_path / "probe.py"
    probe.write_text("import os, sys\nprint(os.environ['RANK'], os.environ['WORLD_SIZE'], os.environ['MASTER_ADDR'], os.environ['LOCAL_RANK'], 'torch' in sys.modules)\n")
    code = ("import sys, pathlib; sys.path.insert(0, %r); import bench; bench.__file__ = %r; "
            "rc = bench.launch_ranks(3, []); print('parent torch', 'torch' in sys.modules); sys.exit(rc)") % (str(ROOT), str(probe))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = sorted(l for l in r.stdout.splitlines() if l and l[0].isdigit())
    assert lines == ["0 3 127.0.0.1 0 False", "1 3 127.0.0.1 1 False", "2 3 127.0.0.1 2 False"]
    assert "parent torch False" in r.stdout


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_through_the_self_launcher():
    """`python bench.py --gpus 2` on a 1-GPU box: two ranks share GPU 0 and exchange gradients over gloo (plumbing check
    of the launcher, the two-graph learner split and the checksum all-reduce; RCCL needs one GPU per rank)."""
    r = run_bench(["--gpus", "2", "--steps", "6", "--warmup", "4", "--envs-per-gpu", "512", "--no-cpu-baseline", "--pretrain-updates", "0", "--steady-steps", "0"],
                  env={"KS_DIST_BACKEND": "gloo", "KS_VISIBLE_GPUS": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "env-shard x2 + grad all-reduce"
    assert line["rccl"]["ranks"] == 2 and line["rccl"]["backend"] == "gloo"
    assert line["replica_weight_checksum_spread"] == 0.0
    assert line["config"]["learner_updates_timed"] == 6 and line["nonfinite_envs"] == 0


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_exchange_gradients_through_peer_mapped_memory():
    """The same two ranks with KS_P2P=1: gloo only carries the rendezvous, the handles and the checksums; the gradients go
    through exchange.PeerExchange (the default under RCCL on a multi-GPU node).  Replicas must stay bit-identical."""
    r = run_bench(["--gpus", "2", "--steps", "4", "--warmup", "3", "--envs-per-gpu", "512", "--no-cpu-baseline", "--pretrain-updates", "0", "--steady-steps", "0"],
                  env={"KS_DIST_BACKEND": "gloo", "KS_VISIBLE_GPUS": "1", "KS_P2P": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl"]["exchange"].startswith("peer-mapped memory"), line["rccl"]
    assert line["replica_weight_checksum_spread"] == 0.0 and line["nonfinite_envs"] == 0


@pytest.mark.gpu
def test_two_ranks_two_gpus_rccl():
    """the real thing when the box has two GPUs: RCCL (backend nccl) over xGMI, replicas stay bit-identical"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    r = run_bench(["--gpus", "2", "--steps", "10", "--warmup", "5", "--no-cpu-baseline", "--pretrain-updates", "0", "--steady-steps", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl"]["backend"] == "nccl" and line["replica_weight_checksum_spread"] == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_peer_exchange_between_processes(world):
    """The LDS-free gradient all-reduce (csrc/ks_xchg.hip, exchange.PeerExchange): `world` processes on this box's GPU(s) map
    each other's exchange blocks through hipIpc handles and run the learner's pattern of all-reduces; every result must be the
    rank-order fp32 mean, bit for bit, and the self-test against the process group's all_reduce must pass."""
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   KS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "tools" / "xchg_selftest.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, out))
    for rc, out in outs:
        assert rc == 0 and "peer exchange OK" in out, out[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("stage", ["connect", "selftest"])
def test_peer_exchange_failure_on_one_rank_makes_every_rank_fall_back(stage):
    """First-contact robustness for the 8-GPU run: if peer access does not work on ONE rank (hipIpcOpenMemHandle fails there, or its
    exchange returns a wrong mean in the connect-time self-test - injected with KS_XCHG_INJECT), ALL ranks must take the same
    decision, keep the process group's all_reduce, say so in the bench line, and still end with bit-identical replicas."""
    r = run_bench(["--gpus", "2", "--steps", "4", "--warmup", "3", "--envs-per-gpu", "512", "--no-cpu-baseline", "--pretrain-updates", "0", "--steady-steps", "0"],
                  env={"KS_DIST_BACKEND": "gloo", "KS_VISIBLE_GPUS": "1", "KS_P2P": "1", "KS_XCHG_INJECT": f"{stage}:1"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    ex = line["rccl"]["exchange"]
    assert ex.startswith("gloo all_reduce (") and "rank" in ex and not ex.startswith("peer-mapped"), ex
    assert line["replica_weight_checksum_spread"] == 0.0 and line["nonfinite_envs"] == 0 and line["rccl"]["exchange_failed_call"] is None


@pytest.mark.gpu
def test_eight_ranks_config5_dry_run_on_one_gpu():
    """`bench.py --gpus 8 --config 5` end to end - the command of BASELINE config 5 on a node - as 8 ranks sharing this box's GPU
    (gloo rendezvous, gradients through the peer exchange, 256 envs per rank): launcher, sharding of scenarios.config5_states by
    rank, mixed-object contexts, the two-graph learner split, periodic replica checks (every 20 updates here) and the final
    checksum all work with world size 8 before the first real node ever sees them."""
    r = run_bench(["--gpus", "8", "--config", "5", "--steps", "6", "--warmup", "4", "--envs-per-gpu", "256", "--no-cpu-baseline",
                   "--pretrain-updates", "45", "--steady-steps", "0"],
                  env={"KS_DIST_BACKEND": "gloo", "KS_VISIBLE_GPUS": "1", "KS_P2P": "1", "KS_REPLICA_CHECK_EVERY": "20"}, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["config"]["envs_per_gpu"] == 256 and "14 README shapes" in line["config"]["workload"]
    assert line["rccl"]["ranks"] == 8 and line["rccl"]["exchange"].startswith("peer-mapped memory"), line["rccl"]
    # (free-running ranks check their replicas at the END of a rollout launch that crossed a multiple of KS_REPLICA_CHECK_EVERY, not
    # from inside the update path - the persistent kernel holds the LDS the check's collectives need: one 45-step launch = one check)
    assert line["replica_weight_checksum_spread"] == 0.0 and line["replica_checks_during_run"] >= 1 and line["nonfinite_envs"] == 0


@pytest.mark.gpu
def test_free_running_rollout_bench_line():
    """`bench.py --rollout free`: the persistent rollout kernel + learner graphs through the whole bench protocol (priming without
    learning, pre-training, warm-up, K timed steps = K learner updates, steady window), no dropped episode, finite."""
    r = run_bench(["--rollout", "free", "--chunk", "5", "--steps", "10", "--warmup", "5", "--envs-per-gpu", "512", "--no-cpu-baseline",
                   "--pretrain-updates", "40", "--steady-steps", "30"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["launch"].startswith("free-running rollout kernel") and line["roofline"]["kernel"].startswith("k_rollout")
    fr = line["config"]["free_running"]
    assert fr["episodes_dropped"] == 0 and fr["episodes_finished"] >= 512 and fr["episodes_kept"] > 0
    assert line["config"]["learner_updates_timed"] == 10 and line["nonfinite_envs"] == 0 and line["steps"] == 10
    assert line["steady_state"]["learner_updates_timed"] == 30 and line["roofline"]["avg_launch_ms"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("p2p", ["1", "0"])
def test_two_ranks_free_running_rollout(p2p):
    """`bench.py --gpus 2 --rollout free` (two ranks sharing this box's GPU): each rank's persistent rollout kernel runs its own env shard,
    the learners' updates meet in the gradient all-reduces (peer exchange with KS_P2P=1, the process group's all_reduce with 0) - the replicas
    must end bit-identical although every rank's rollout is free-running, and no episode may be dropped."""
    r = run_bench(["--gpus", "2", "--rollout", "free", "--chunk", "5", "--steps", "10", "--warmup", "5", "--envs-per-gpu", "512", "--no-cpu-baseline",
                   "--pretrain-updates", "45", "--steady-steps", "0"], env={"KS_DIST_BACKEND": "gloo", "KS_VISIBLE_GPUS": "1", "KS_P2P": p2p, "KS_REPLICA_CHECK_EVERY": "20"})
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["launch"].startswith("free-running rollout kernel")
    assert line["replica_weight_checksum_spread"] == 0.0 and line["replica_checks_during_run"] >= 2 and line["nonfinite_envs"] == 0
    assert line["config"]["free_running"]["episodes_dropped"] == 0 and line["config"]["learner_updates_timed"] == 10
    assert line["rccl"]["exchange"].startswith("peer-mapped memory" if p2p == "1" else "gloo all_reduce")


@pytest.mark.gpu
def test_two_ranks_free_running_rollout_with_replicas_averaged_per_launch():
    """The node-run fallback of round 5: when the ranks cannot map each other's memory and the process group's
    collectives are library kernels (RCCL), AsyncTrainer keeps the free-running rollout, applies each launch's updates with LOCAL gradients
    and averages parameters / targets / Adam moments at the launch boundary (KS_ASYNC_SYNC=average forces that mode here, over gloo, two
    ranks sharing this box's GPU).  The ranks' shards differ, so their local updates differ - yet the replicas must be bit-identical
    after every launch, the line must say which synchronisation ran, and no episode may be dropped."""
    r = run_bench(["--gpus", "2", "--rollout", "free", "--chunk", "5", "--steps", "10", "--warmup", "5", "--envs-per-gpu", "512", "--no-cpu-baseline",
                   "--pretrain-updates", "45", "--steady-steps", "0"],
                  env={"KS_DIST_BACKEND": "gloo", "KS_VISIBLE_GPUS": "1", "KS_P2P": "0", "KS_ASYNC_SYNC": "average", "KS_REPLICA_CHECK_EVERY": "20"})
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["launch"].startswith("free-running rollout kernel")
    assert line["rccl"]["replica_sync"] == "average-per-launch" and "replicas averaged per launch" in line["config"]["parallelism"]
    assert "local updates" in line["rccl"]["exchange"]
    assert line["replica_weight_checksum_spread"] == 0.0 and line["replica_checks_during_run"] >= 2 and line["nonfinite_envs"] == 0
    assert line["config"]["free_running"]["episodes_dropped"] == 0 and line["config"]["learner_updates_timed"] == 10


@pytest.mark.gpu
def test_bench_line_on_a_multi_geom_object():
    """`bench.py --shape TBottleS`: the single-object workload on a multi-geom object (libkinova_sim_mg.so) - free-running rollout kernel
    with the learner beside it; the line names the object, carries no PMC block of another workload, and no env raised a status flag"""
    r = run_bench(["--shape", "TBottleS", "--steps", "20", "--warmup", "10", "--init-policy", "none", "--pretrain-updates", "60", "--steady-steps", "30", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert "TBottleS" in d["config"]["workload"] and "5 cm correction" in d["config"]["reset"]
    assert d["config"]["free_running"] is not None and d["config"]["free_running"]["episodes_dropped"] == 0
    assert d["roofline"]["traffic"] is None and d["value"] > 1e6
    assert all(v == 0 for v in d["status_counts"].values()) and d["nonfinite_envs"] == 0
