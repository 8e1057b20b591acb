"""Pipeline for rollout + DDPGfD training on one GPU (one process per GPU): HIP graphs for the many-kernel parts,
direct launches for the one-kernel parts, the learner on a second stream beside the simulator.

Per env-step, on the main stream (the critical chain):

    pre      action selection: ONE kernel (kr_actor_select = actor forward on MFMA + in-kernel exploration noise +
             check_grasp latch / scripted lift), launched directly; a graph (g_pre) when the actor is not a supported MLP
    ks_step  the simulator (libkinova_sim): k_env_step (15 substeps + the rays of the workgroup's envs) and k_obs
    post     kr_store_transition (open-episode buffers + per-env bookkeeping), launched directly

and on the learner's stream, started right behind the action selection:

    g_commit finished episodes of the PREVIOUS step enter the replay ring (rank / commit / advance)
    g_head   actor Adam + soft target update (previous update's gradients), window sampling
    g_learn  the body of one DDPGfD update (learner_native); with world_size > 1 it is two graphs (critic backward |
             critic step + actor backward) with the all-reduce of the flat gradient buffers (RCCL) issued after each -
             collectives stay outside the captures.

The learner is a one-step software pipeline: each update is a short HEAD (the actor's Adam step + soft target update
for the gradients of the PREVIOUS update, then the window sampling) and a BODY (targets, critic backward + Adam, actor
backward).  The rollout only waits for the head before it writes the replay.  At 256-256 every launch of the update is
LDS-free (csrc/ks_mlp.hip), so the whole body executes on the registers, matrix pipes and issue slots the stepping
kernel leaves idle and ends before that kernel does; with library GEMMs (other widths) the body can only get CUs once
simulator workgroups retire (the stepping kernel holds all of every CU's LDS) and runs behind it.
The actor therefore acts with weights that lag the learner by one update; the reference's own loop acts with a policy
that is a whole episode old (100 updates at the end of each episode, main_DDPGfD.py:466-486).
"""
from __future__ import annotations

import os

import torch


def _concurrent_stream(main, dev, tries=8):
    """A second stream whose work really runs BESIDE `main`'s.  Streams share the GPU's few hardware queues (4 by default,
    GPU_MAX_HW_QUEUES) and torch hands out pooled streams round-robin: in a process that has created streams before, a new stream
    can land on `main`'s queue - its kernels then start only when `main`'s have finished, the learner would run BEHIND the stepping
    kernel instead of beside it and the free-running rollout would drop the episodes the learner's stream does not collect in time.
    Probe: a spin kernel on `main`, an event behind a trivial op on the candidate; returns (stream, overlapped)."""
    import time
    import warnings
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(main):
        e0.record(main)
        torch.cuda._sleep(1000000)
        e1.record(main)
    torch.cuda.synchronize(dev)
    ms = max(1e-3, e0.elapsed_time(e1))
    spin = int(1000000 * min(1e4, 25.0 / ms))               # ~25 ms
    flag = torch.zeros(1, device=dev)
    cand = None
    for _ in range(tries):
        cand = torch.cuda.Stream(dev)
        ev = torch.cuda.Event()
        with torch.cuda.stream(main):
            torch.cuda._sleep(spin)
        with torch.cuda.stream(cand):
            flag.add_(1.0)
            ev.record(cand)
        t0 = time.perf_counter()
        while not ev.query() and time.perf_counter() - t0 < 0.010:
            pass
        ok = ev.query()
        torch.cuda.synchronize(dev)
        if ok:
            return cand, True
    warnings.warn("no stream found whose work overlaps the current stream's (GPU_MAX_HW_QUEUES too small?): the learner will run behind the stepping "
                  "kernels, not beside them", RuntimeWarning)
    return cand, False


class GraphedTrainer:
    def __init__(self, sim, policy, replay, engine, batch_episodes=64, overlap=True, learn_after=31, expert_replay=None, expert_prob=0.3,
                 updates_per_step=1):
        """updates_per_step U / batch_episodes: the update-to-data knobs.  The reference makes 100 updates per 30-step episode of ONE
        env (main_DDPGfD.py:474-486: ~3 per stored transition); BASELINE config 3 - bench.py's workload - is ONE update on 64 episodes
        per env-step of 4096 envs.  U > 1 replays the captured update U times per env-step (the step becomes learner-bound beyond
        U = 1); a larger batch is nearly free instead: the LDS-free learner kernels are one wave per 16 rows and latency-bound, so as
        long as a pass fits the SIMDs' spare wave slots beside the stepping kernel (~1000 workgroups = 640 episodes) its duration
        hardly grows with the rows.
        expert_replay (a DeviceEpisodeReplay filled by demonstrators.run_controller_episodes or loaded from a reference replay
        bundle) turns the update into DDPGfD proper: every batch is int(batch_episodes * (1 - expert_prob)) agent episodes + the rest
        expert episodes (DDPGfD.train_batch, DDPGfD.py:232-254), sampled by ONE launch inside the captured update."""
        assert engine.gen is None, "graph capture uses the default CUDA generator"
        self.sim, self.policy, self.replay, self.eng = sim, policy, replay, engine
        self.expert_replay, self.expert_prob = expert_replay, float(expert_prob)
        if expert_replay is not None and not (replay.native and expert_replay.native):
            raise ValueError("GraphedTrainer: the expert mix needs device rings (DeviceEpisodeReplay on the GPU)")
        self.batch_episodes, self.overlap, self.learn_after = batch_episodes, overlap, learn_after
        self.updates_per_step = max(1, int(updates_per_step))
        self.dev = sim.device
        self.steps = 0
        self.updates = 0
        self.distributed = False
        try:
            import torch.distributed as dist
            self.distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(policy.process_group) > 1
        except Exception:
            pass
        self.main = torch.cuda.current_stream(self.dev)
        self.side, self.streams_overlap = _concurrent_stream(self.main, self.dev) if overlap else (torch.cuda.Stream(self.dev), None)
        self.acted = torch.cuda.Event()
        self.head_done = torch.cuda.Event()
        self.g_pre = self.g_post = self.g_commit = None
        self.defer_commit = self.pending_commit = False
        self.direct_pre = self.direct_post = False
        self.g_learn = []
        self.losses = None
        # the update itself: explicit GEMMs + fused glue kernels (learner_native), same arithmetic as policy.train_on_batch
        from .learner_native import NativeDDPGfDUpdate
        self.native = NativeDDPGfDUpdate(policy)
        self.native.pipelined = True
        # the gradient exchange between the ranks: an LDS-free all-reduce over peer-mapped memory where the ranks can map each
        # other's buffers (one process per GPU on a node) - it runs beside the stepping kernel, which a library collective
        # cannot (its kernels need LDS).  Self-tested against the process group at start-up; KS_P2P=0 keeps the library path.
        self.exchange_note = "none (single rank)"
        if self.distributed:
            import torch.distributed as dist
            backend = dist.get_backend(policy.process_group)
            want = os.environ.get("KS_P2P", "1" if backend == "nccl" else "0") != "0"
            if want:
                from .exchange import try_peer_exchange
                n_max = max(self.native.actor.flat.numel(), self.native.critic.flat.numel())
                self.native.exchange, why = try_peer_exchange(n_max, policy.process_group, self.dev)
                self.exchange_note = "peer-mapped memory, LDS-free (ks_xchg)" if why is None else f"{backend} all_reduce ({why})"
            else:
                self.exchange_note = f"{backend} all_reduce"
        self.sample_seed = int(torch.initial_seed())          # the window sampler's Philox key
        # first-contact safety on a real node: every `check_every` updates the exchange's error word is read (a rank whose wait for a
        # peer ran out stops reducing: fail loudly instead of training diverged replicas) and the replicas' weight checksums are
        # compared over the process group.  Both are host reads, hence not every update.
        import os as _os
        self.check_every = int(_os.environ.get("KS_REPLICA_CHECK_EVERY", "500")) if self.distributed else 0
        self.replica_checks = 0
        self.check_in_body = True              # AsyncTrainer checks at the end of a launch instead (its rollout kernel holds every CU's LDS)

    # -- learner phases on the static batch -------------------------------------------------------------
    def _sample(self):
        # (uniforms drawn in the sampling kernel, keyed by the update count: no generator-state launches in the graph)
        if self.expert_replay is not None and self.expert_prob > 0:
            self.batch = self.replay.sample_mixed(self.expert_replay, self.batch_episodes, self.expert_prob, draw=self.native.it, seed=self.sample_seed)
        else:
            self.batch = self.replay.sample_batch_nstep(self.batch_episodes, draw=self.native.it if self.replay.native else None, seed=self.sample_seed)

    def _head(self):
        self.native.phase_head()
        self._sample()

    def _phase1(self):
        st, ac, ns, rw, nd, w = self.batch[:6]
        self.loss_c = self.native.phase_critic(st, ac, ns, rw, w, next_ends=self.batch[6] if len(self.batch) > 6 else None)

    def _phase2(self):
        self.native.phase_actor(self.batch[0], self.batch[5])     # (its actor Adam step runs in the next update's head: native.pipelined)

    def _learn_eager(self):
        self._head()
        self._phase1()
        self.native.allreduce("critic")
        self._phase2()
        self.native.allreduce("actor")

    def capture(self, warmup_steps=3):
        """Run `warmup_steps` eager steps (allocator / autotune warm-up, as torch.cuda.graphs requires) and capture."""
        eng, sim = self.eng, self.sim
        # The learner's GEMMs are small and skinny (M = 1600 / 8000 rows, N, K <= 256): let PyTorch's TunableOp time the
        # hipBLASLt / rocBLAS candidates for each shape during the eager warm-up and keep the fastest; tuning is
        # switched off again before the captures (the selections stay in use).  KS_TUNABLEOP=0 skips it.
        tune = os.environ.get("KS_TUNABLEOP", "1") != "0" and hasattr(torch.cuda, "tunable")
        if tune:
            torch.cuda.tunable.enable(True)
            torch.cuda.tunable.tuning_enable(True)
            torch.cuda.tunable.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), "ks_tunableop.csv"), insert_device_ordinal=True)
            torch.cuda.tunable.set_max_tuning_duration(15)
        # The eager warm-up runs real learner updates - on an EMPTY replay when capture() is called at the start of training
        # (all-padding batches: zero gradients, but Adam's weight decay and bias correction would still move the weights
        # and advance the counters).  The learner's whole state is therefore saved here and restored after the warm-up: the
        # warm-up only warms the allocator and the GEMM selection, it never trains.
        nat, pol = self.native, self.policy
        saved = {k: v.clone() for k, v in pol._flat_params.items()}
        saved_opt = [(net, net.grad.clone(), net.exp_avg.clone(), net.exp_avg_sq.clone()) for net in (nat.actor, nat.critic)]
        saved_it, saved_head, saved_total = nat.it.clone(), nat.it_head.clone(), pol.total_it
        s = torch.cuda.Stream(self.dev)
        s.wait_stream(self.main)
        with torch.cuda.stream(s):
            for _ in range(warmup_steps):
                eng.pre()
                sim.step(eng.action_t)
                eng.post()
                self._learn_eager()
                self.steps += 1
        self.main.wait_stream(s)
        torch.cuda.synchronize(self.dev)
        for k, v in saved.items():
            pol._flat_params[k].copy_(v)
        for net, g, m, v in saved_opt:
            net.grad.copy_(g); net.exp_avg.copy_(m); net.exp_avg_sq.copy_(v)
        nat.it.copy_(saved_it); nat.it_head.copy_(saved_head)
        pol.total_it = saved_total
        if tune:
            torch.cuda.tunable.tuning_enable(False)
        # thread_local: other threads (the RCCL watchdog of torch.distributed) keep issuing HIP calls during a capture
        mode = dict(capture_error_mode="thread_local")
        # Action selection and the replay write are ONE kernel each on the fused path (kr_actor_select with its in-kernel
        # noise; kr_store_transition): those are launched directly - a one-node graph only adds the graph -> stream
        # hand-over (~10-20 us on this runtime) in front of the stepping kernel.
        self.direct_pre = eng.native and eng.device_noise and eng._fused_actor_layers() is not None
        if not self.direct_pre:
            self.g_pre = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_pre, **mode):
                eng.pre()
        # With the learner on its own stream the ring update of a step (rank / commit / advance) is deferred to the start of
        # the NEXT step on that stream, ahead of the window sampling that needs it: three launches less between two
        # launches of the stepping kernel.
        self.defer_commit = self.overlap and eng.native and eng.replay is not None
        self.direct_post = self.defer_commit
        if not self.direct_post:
            self.g_post = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_post, **mode):
                eng.post(commit=not self.defer_commit)
        if self.defer_commit:
            self.g_commit = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_commit, **mode):
                eng.commit()
        self.g_head = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_head, **mode):
            self._head()
        phases = [self._phase1, self._phase2]
        groups = [[p] for p in phases] if self.distributed else [phases]
        pool = self.g_head.pool()
        for grp in groups:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, **mode):
                for p in grp:
                    p()
            pool = g.pool()
            self.g_learn.append(g)
        torch.cuda.synchronize(self.dev)

    def _body(self):
        if not self.distributed:
            self.g_learn[0].replay()
        else:
            self.g_learn[0].replay()
            self.native.allreduce("critic")
            self.g_learn[1].replay()
            self.native.allreduce("actor")
        self.updates += 1
        if self.check_every and self.check_in_body and self.updates % self.check_every == 0:
            self.check_replicas()

    def replica_checksum_spread(self) -> float:
        """max over the four networks of |max - min| over the ranks of two checksums (sum, sum of |.|) of the flat parameters: 0.0
        when the replicas are bit-identical.  Two small collectives on the process group + a host read."""
        import torch.distributed as dist
        pol = self.policy
        chk = torch.stack([f(pol._flat_params[k].double()) for k in ("actor", "critic", "actor_target", "critic_target")
                           for f in (torch.sum, lambda x: x.abs().sum())])
        hi, lo = chk.clone(), chk.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=pol.process_group)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=pol.process_group)
        return float((hi - lo).abs().max().item())

    def check_replicas(self):
        """Called every `check_every` updates on the learner's stream position (all ranks reach it at the same update count)."""
        if not self.distributed:
            return
        cur = torch.cuda.current_stream(self.dev)
        cur.wait_stream(self.side)
        if self.native.exchange is not None:
            self.native.exchange.check()
        spread = self.replica_checksum_spread()
        self.replica_checks += 1
        if spread != 0.0:
            raise RuntimeError(f"replica weights differ between ranks after {self.updates} updates (checksum spread {spread:.3e}); exchange: {self.exchange_note}")

    def step(self):
        """One env-step for every env + one learner update (once the replay holds episodes)."""
        main, side = self.main, self.side
        if self.direct_pre:
            self.eng.pre()
        else:
            self.g_pre.replay()
        learn = self.steps >= self.learn_after
        if self.overlap and (learn or self.pending_commit):
            self.acted.record(main)
        self.sim.step(self.eng.action_t)
        if self.overlap:
            if learn or self.pending_commit:
                side.wait_event(self.acted)        # the actor's weights are free once this step's forward is done
                with torch.cuda.stream(side):
                    self._commit_pending()         # episodes that finished in the previous step enter the ring
                    if learn:
                        self.g_head.replay()
                    self.head_done.record(side)
                    if learn:
                        self._body()
                        for _ in range(self.updates_per_step - 1):       # further updates of this env-step: behind the first, beside the simulator
                            self.g_head.replay()
                            self._body()
                main.wait_event(self.head_done)    # ring updated, windows sampled, actor weights settled: the body runs on its own
        elif learn:
            for _ in range(self.updates_per_step):
                self.g_head.replay()
                self._body()
        if self.direct_post:
            self.eng.post(commit=False)
        else:
            self.g_post.replay()
        self.pending_commit = self.defer_commit
        self.steps += 1
        return self.eng.reward_out, self.eng.done_out

    def _commit_pending(self):
        if self.pending_commit:
            self.g_commit.replay()
            self.pending_commit = False

    def flush(self, finish_update=False):
        """Apply the deferred ring update of the last step (call before reading the replay from the host).
        finish_update=True also applies the last update's pending actor step (call before saving a checkpoint; DDPGfD.save
        does it by itself through the native learner)."""
        if self.pending_commit or finish_update:
            self.side.wait_stream(self.main)
            with torch.cuda.stream(self.side):
                self._commit_pending()
                if finish_update:
                    self.native.finish_pending()
            self.main.wait_stream(self.side)
        if self.distributed and self.native.exchange is not None:
            self.native.exchange.check()


class AsyncTrainer(GraphedTrainer):
    """Free-running rollout + learner (single GPU or one rank of several): the whole rollout side of an env-step - actor forward,
    exploration noise, check_grasp / scripted lift, the 15 substeps, rays, observation, replay write - is ONE persistent launch
    (ks_rollout) in which every stepping workgroup loops over its own 16 envs without waiting for any other workgroup, while the
    learner's captured update graphs run beside it on a second stream.

    Why: a lock-step launch lasts as long as its slowest wave (1.6 - 1.9 x the median wave) and the ~60 us of
    launch gaps / actor / replay-write kernels between two launches idle every CU; free-running, a CU starts its next env-step
    the moment it has finished the last.  What changes semantically: envs no longer advance in lock step (after K steps every env
    has done K steps, but at different wall-clock times), the actor weights an env acts with are the newest PUBLISHED ones
    (triple-buffered, published after every actor Adam step) instead of exactly one update old, and finished episodes reach the
    replay ring in arrival order - training is no longer bit-reproducible run to run (GraphedTrainer remains the reproducible
    path).  Per env the arithmetic and the noise stream are the lock-step ones (tests/test_gpu_async.py: identical trajectories
    for fixed weights).  The reference's own loop acts with a policy that is a whole episode old (main_DDPGfD.py:466-486)."""

    def __init__(self, sim, policy, replay, engine, batch_episodes=64, expert_replay=None, expert_prob=0.3, updates_per_step=1):
        super().__init__(sim, policy, replay, engine, batch_episodes=batch_episodes, overlap=True, expert_replay=expert_replay, expert_prob=expert_prob,
                         updates_per_step=updates_per_step)
        from .sim import KsRolloutArgs
        eng, dev = engine, self.dev
        if not self.native.lds_free:
            raise ValueError("AsyncTrainer needs the LDS-free learner kernels (hidden widths 256-256 / 128-128 / 64-64): the persistent rollout kernel "
                             "holds every CU's LDS for the whole launch, a learner built on library GEMMs could only run behind it")
        # How the replicas are kept together:
        #   "per-update"         every update's two gradient buffers are averaged over the ranks before Adam (north_star's scheme).  Beside the
        #                        persistent kernel that is the LDS-free peer exchange (or a host-side backend: gloo).  When the ranks cannot map each
        #                        other's memory under nccl, the collectives are library KERNELS (RCCL) that need LDS - which the persistent rollout
        #                        kernel holds on every CU until its launch ends: the learner's stream then only advances BETWEEN launches.  The
        #                        algorithm stays north_star's; callers bound the learner's lag by keeping launches short (`library_allreduce_chunk`
        #                        env-steps: bench.py lowers --chunk to it).  The default since round 6.
        #   "average-per-launch" opt-in (KS_ASYNC_SYNC=average): each rank applies the updates of a launch with its LOCAL gradients beside its own
        #                        rollout, and at the END of the launch parameters, target parameters and Adam moments are averaged over the ranks
        #                        (one all-reduce of the flat buffers) and the averaged actor is published.  Periodic model averaging / local SGD: a
        #                        DIFFERENT algorithm from the per-update gradient all-reduce - it warns when chosen and the bench line names it.
        self.replica_sync = "per-update" if self.distributed else "single rank"
        self.library_allreduce_chunk = None
        if self.distributed and self.native.exchange is None:
            import torch.distributed as dist
            forced = os.environ.get("KS_ASYNC_SYNC", "")
            if forced == "average":
                import warnings
                warnings.warn("AsyncTrainer: KS_ASYNC_SYNC=average - replicas apply LOCAL updates and are averaged at every launch boundary (periodic model "
                              "averaging), not north_star's per-update gradient all-reduce", RuntimeWarning)
                self.replica_sync = "average-per-launch"
                self.native.local_gradients = True                       # native.allreduce() becomes a no-op: the body applies local gradients
                self.exchange_note += "; free-running rollout kept: local updates, replicas averaged at every launch boundary"
            elif dist.get_backend(policy.process_group) == "nccl":
                self.library_allreduce_chunk = int(os.environ.get("KS_ASYNC_LIBRARY_CHUNK", "6"))
                self.exchange_note += (f"; no peer mapping: per-update library all-reduce, which executes between launches - keep launches at <= "
                                       f"{self.library_allreduce_chunk} env-steps (the learner lags by at most one launch)")
        if not (eng.native and eng.device_noise and eng._fused_actor_layers() is not None and sim.cfg.auto_reset and sim.obs_env_major):
            raise ValueError("AsyncTrainer needs the fused actor path (3-layer MLP at a supported width, in-kernel noise), auto_reset and env-major obs")
        flat = policy._flat_params["actor"]
        actor = policy.actor
        off = lambda t: (t.data_ptr() - flat.data_ptr()) // 4
        self.actor_flat = flat
        stride = (flat.numel() + 3) // 4 * 4
        self.pub = torch.zeros(3, stride, device=dev)
        self.pub_ver = torch.zeros(1, dtype=torch.long, device=dev)
        self.n_pub = 0
        self.pub[0, :flat.numel()].copy_(flat)
        replay.enable_async()
        self.steps_total = torch.zeros(eng.n, dtype=torch.long, device=dev)
        self.counters = torch.zeros(8 + 4 * 512 + 8 + (eng.n if os.environ.get("KS_DEBUG_DROPS") else 0), dtype=torch.long, device=dev)      # episodes finished, lifted, kept, dropped (+ 4 phase timers and 4 x 512
                                                                          # per-workgroup stamps of the -DKS_ROLLOUT_STAMP diagnostic build)
        P = lambda t: t.data_ptr()
        a = KsRolloutArgs()
        a.actor_pub, a.actor_ver, a.actor_stride = P(self.pub), P(self.pub_ver), stride
        a.off_w1, a.off_b1, a.off_w2, a.off_b2, a.off_w3, a.off_b3 = (off(actor.l1.weight), off(actor.l1.bias), off(actor.l2.weight), off(actor.l2.bias),
                                                                      off(actor.l3.weight), off(actor.l3.bias))
        a.h1, a.h2 = actor.l1.weight.shape[0], actor.l2.weight.shape[0]
        from .rollout import SKIP_NUM_TS
        a.sigma, a.max_action, a.skip_steps, a.with_replay, a.seed = eng.sigma, eng.max_action, SKIP_NUM_TS, 1, eng.noise_seed
        a.obs, a.prev_obs, a.has_prev, a.ready, a.lifting = P(eng.obs), P(eng.prev_obs), P(eng.has_prev), P(eng.ready), P(eng.lifting)
        a.t, a.steps_total, a.action, a.action_t = P(eng.t), P(self.steps_total), P(eng.action), P(eng.action_t)
        a.reward_out, a.done_out = P(eng.reward_out), P(eng.done_out)
        a.sim_obs, a.sim_reward, a.sim_done, a.sim_info, a.sim_final_obs = P(sim.obs), P(sim.reward), P(sim.done), P(sim.info), P(sim.final_obs)
        a.horizon, a.n_steps = replay.horizon, replay.n_steps
        a.cur_state, a.cur_next, a.cur_action = P(replay.a_state), P(replay.a_next), P(replay.a_action)
        a.cur_reward, a.cur_not_done, a.cur_len = P(replay.a_reward), P(replay.a_not_done), P(replay.a_len)
        a.cur_sel, a.pub_len, a.counters = P(replay.a_sel), P(replay.pub_len), P(self.counters)
        # pacing: the learner's stream waits (kr_wait_min on the envs' step counters) so that update k of a launch starts when EVERY env
        # has done k - lead env-steps.  Without it the learner (0.9 ms per update) finishes its share of a long launch far ahead of the
        # rollout (1.1 ms per env-step) and nobody collects the episodes published after that.
        from .sim import load_library
        self._lib = load_library()
        self.check_in_body = False
        self.pace_lead = int(os.environ.get("KS_ASYNC_LEAD", "8"))           # < 0: no pacing
        self.pace_timeouts = torch.zeros(1, dtype=torch.long, device=dev)    # waits of the pacing kernel that ended on the clock (counts())
        self.args = a
        self.env_steps = 0
        # The persistent launch has one workgroup per compute unit at most; with more 16-env groups than CUs (BASELINE config 5: 8192 envs)
        # every workgroup steps two or three groups in turn (k_rollout).  Balanced only when the groups divide evenly over the CUs: a
        # workgroup with one group more than the others sets the launch's pace - say so.
        self.rollout_plan, n_groups, n_wgs = sim.rollout_plan()              # the library's own decision (ks_rollout_plan)
        self.groups_per_workgroup = (n_groups + n_wgs - 1) // n_wgs
        if self.rollout_plan in ("runs", "round-robin") and n_groups % n_wgs:
            import warnings
            warnings.warn(f"AsyncTrainer: {n_groups} env groups dealt to {n_wgs} persistent workgroups ({self.rollout_plan}): some step {self.groups_per_workgroup} groups per "
                          f"env-step, others {self.groups_per_workgroup - 1} - the launch runs at the pace of the former (the ready queue, KS_ROLLOUT_DEAL=queue, and "
                          "GraphedTrainer's lock-step launches balance such a batch dynamically)", RuntimeWarning)

    def publish(self):
        """make the actor's current weights the newest published version: copy into the buffer two behind the one in use, then
        advance the version counter (stream order: the copy is complete - and written back at its kernel's end - before the counter
        moves; the rollout workgroups read the counter with an agent-scope load at the start of each of their env-steps)"""
        self.n_pub += 1
        self.pub[self.n_pub % 3, :self.actor_flat.numel()].copy_(self.actor_flat)
        self.pub_ver.fill_(self.n_pub)

    def capture(self, warmup_updates=2):
        """captures the learner's update (head + body); the rollout is a single launch and needs no graph"""
        # the learner's LDS-free kernels must fit beside the persistent kernel's waves: 368 of the 512 registers per lane are taken
        # for the whole launch, the 4-wave split forward / backward (160) does not fit, the one-wave variants (128) do - the choice
        # (mlp._split_waves reads KS_MLP_SPLIT at every call) is baked into the captured graphs
        before = os.environ.get("KS_MLP_SPLIT")
        os.environ["KS_MLP_SPLIT"] = os.environ.get("KS_ASYNC_MLP_SPLIT", "0")      # (experiment: a build whose split kernels fit in 128 registers)
        try:
            self._capture(warmup_updates)
        finally:
            if before is None:
                os.environ.pop("KS_MLP_SPLIT", None)
            else:
                os.environ["KS_MLP_SPLIT"] = before

    def _capture(self, warmup_updates):
        nat, pol = self.native, self.policy
        saved = {k: v.clone() for k, v in pol._flat_params.items()}
        saved_opt = [(net, net.grad.clone(), net.exp_avg.clone(), net.exp_avg_sq.clone()) for net in (nat.actor, nat.critic)]
        saved_it, saved_head, saved_total = nat.it.clone(), nat.it_head.clone(), pol.total_it
        s = torch.cuda.Stream(self.dev)
        s.wait_stream(self.main)
        with torch.cuda.stream(s):
            for _ in range(warmup_updates):
                self.replay.commit_published()
                self._learn_eager()
        self.main.wait_stream(s)
        torch.cuda.synchronize(self.dev)
        for k, v in saved.items():
            pol._flat_params[k].copy_(v)
        for net, g, m, v in saved_opt:
            net.grad.copy_(g); net.exp_avg.copy_(m); net.exp_avg_sq.copy_(v)
        nat.it.copy_(saved_it); nat.it_head.copy_(saved_head)
        pol.total_it = saved_total
        mode = dict(capture_error_mode="thread_local")
        self.g_commit = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_commit, **mode):
            self.replay.commit_published()
        self.g_head = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_head, pool=self.g_commit.pool(), **mode):
            self._head()
        phases = [self._phase1, self._phase2]
        groups = [[p] for p in phases] if self.distributed else [phases]
        pool = self.g_head.pool()
        for grp in groups:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, **mode):
                for p in grp:
                    p()
            pool = g.pool()
            self.g_learn.append(g)
        torch.cuda.synchronize(self.dev)

    def run(self, n_steps: int, learn: bool = True, budget_ms: float | None = None, updates: int | None = None):
        """n_steps env-steps of EVERY env (one persistent launch on the main stream) and, beside it on the learner's stream, n_steps
        updates (`updates` overrides their number).
        budget_ms (opt-in, round 6; needs the wave form of the kernel, ks_rollout_plan): a TIME budget for the launch instead - every wave steps
        its four envs until the budget has passed (at most n_steps env-steps, at least one), so envs advance by time and `steps_total` says how far
        each got.  For contexts whose objects differ widely in cost (a curriculum stage with bowls and cubes: the slowest object otherwise paces
        every env); what is collected then holds more transitions of the cheap objects.  The learner is no longer paced on the envs' step counters
        from the first budgeted launch on.
        Beside the launch: [episodes published so far -> ring, actor step + targets + sampling, publish the actor, body].  Returns after
        ENQUEUEING both; synchronise (or call again) to wait.  The two streams only meet at the start of the next run().
        The learner's stream is the faster one; it is paced on the envs' step counters (update k starts when every env has done
        k - KS_ASYNC_LEAD env-steps, default 8), so published episodes keep being collected until a launch of any length
        ends (an env that finishes more episodes than it has open buffers (2) while nobody collects drops them:
        counts()["episodes_dropped"])."""
        if self.g_commit is None:
            raise RuntimeError("AsyncTrainer.run before capture(): the learner's graphs do not exist yet")
        main, side = self.main, self.side
        side.wait_stream(main)
        main.wait_stream(side)
        if budget_ms is not None:
            self.args.budget_ticks = max(1, int(budget_ms * 1e5))           # 100 MHz device clock
            self.pace_lead = -1                                             # the envs' counters are uneven from here on
        try:
            self.sim.rollout(n_steps, self.args)
        finally:
            self.args.budget_ticks = 0
        n, done = self.eng.n, self.env_steps
        with torch.cuda.stream(side):
            for k in range(n_steps if updates is None else updates):
                if self.pace_lead >= 0 and k > self.pace_lead and (k - self.pace_lead) % 4 == 1:      # (every 4th update: the lead varies between 8 and 11)
                    self._lib.kr_wait_min_counted(self.steps_total.data_ptr(), n, done + k - self.pace_lead, 5.0, self.pace_timeouts.data_ptr(),
                                                  torch.cuda.current_stream(self.dev).cuda_stream)
                self.g_commit.replay()
                if learn:
                    for _ in range(self.updates_per_step):
                        self.g_head.replay()
                        self.publish()
                        self._body()
        self.env_steps += n_steps
        if learn and self.replica_sync == "average-per-launch":
            with torch.cuda.stream(side):
                self.average_replicas()
        # the replica check (process-group collectives + a host read: kernels that need LDS, which the persistent launch holds) runs
        # at the end of a launch, not from inside the update path
        if learn and self.check_every and self.updates // self.check_every != (self.updates - n_steps * self.updates_per_step) // self.check_every:
            self.check_replicas()

    def average_replicas(self):
        """replica_sync == "average-per-launch": the pending (pipelined) actor step is applied, then the flat parameter buffers of the four
        networks and the Adam moments of actor and critic are averaged over the process group (library collectives on the current stream: they
        execute once the persistent launch has released the CUs' LDS), and the averaged actor becomes the newest published version."""
        import torch.distributed as dist
        nat, pol = self.native, self.policy
        nat.finish_pending()
        world = dist.get_world_size(pol.process_group)
        bufs = [pol._flat_params[k] for k in ("actor", "critic", "actor_target", "critic_target")]
        bufs += [b for net in (nat.actor, nat.critic) for b in (net.exp_avg, net.exp_avg_sq)]
        # ONE all-reduce: the eight flat buffers packed into one (5.6 MB at 256-256), averaged, unpacked (it was eight blocking
        # all-reduces and eight scaling kernels per launch boundary)
        flat = torch.cat([b.reshape(-1) for b in bufs])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=pol.process_group)
        flat.div_(world)
        off = 0
        for b in bufs:
            b.copy_(flat[off:off + b.numel()].view_as(b))
            off += b.numel()
        self.publish()

    def step(self):
        self.run(1)
        return self.eng.reward_out, self.eng.done_out

    def flush(self, finish_update=False):
        self.side.wait_stream(self.main)
        with torch.cuda.stream(self.side):
            self.g_commit.replay() if self.g_commit is not None else self.replay.commit_published()
            if finish_update:
                self.native.finish_pending()
                self.publish()
        self.main.wait_stream(self.side)
        if self.distributed and self.native.exchange is not None:
            self.native.exchange.check()

    def counts(self):
        c = self.counters[:4].tolist()
        return {"episodes_finished": c[0], "lifted": c[1], "episodes_kept": c[2], "episodes_dropped": c[3], "pacing_timeouts": int(self.pace_timeouts.item())}
