"""HIP-graph pipeline for rollout + DDPGfD training on one GPU (one process per GPU).

An env-step of the eager RolloutEngine + learner is ~250 small kernels (actor MLP, masked replay writes, window
gather, two backward passes, Adam, target update); launched one by one they cost more host time than the
physics kernel takes on the device.  GraphedTrainer captures them once into HIP graphs and replays them:

    g_pre    action selection (check_grasp, actor forward, exploration noise, scripted lift)
    ks_step  the simulator (libkinova_sim, launched directly on the stream - not part of a graph)
    g_post   replay writes + episode bookkeeping
    g_learn  window sampling + one DDPGfD update (learner_native: explicit GEMMs + fused glue kernels); with
             world_size > 1 it is three graphs (critic backward | critic step + actor backward | actor step +
             targets) with the all-reduce of the flat gradient buffer (RCCL) issued between them - collectives stay
             outside the captures.

The learner graphs run on a second stream beside the simulator kernel: they read the replay as it was before this
step's writes and update the weights after this step's actor forward.
"""
from __future__ import annotations

import torch


class GraphedTrainer:
    def __init__(self, sim, policy, replay, engine, batch_episodes=64, overlap=True, learn_after=31):
        assert engine.gen is None, "graph capture uses the default CUDA generator"
        self.sim, self.policy, self.replay, self.eng = sim, policy, replay, engine
        self.batch_episodes, self.overlap, self.learn_after = batch_episodes, overlap, learn_after
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
        self.side = torch.cuda.Stream(self.dev)
        self.acted = torch.cuda.Event()
        self.g_pre = self.g_post = None
        self.g_learn = []
        self.losses = None
        # the update itself: explicit GEMMs + fused glue kernels (learner_native), same arithmetic as policy.train_on_batch
        from .learner_native import NativeDDPGfDUpdate
        self.native = NativeDDPGfDUpdate(policy)

    # -- learner phases on the static batch -------------------------------------------------------------
    def _sample(self):
        self.batch = self.replay.sample_batch_nstep(self.batch_episodes)

    def _phase1(self):
        self._sample()
        st, ac, ns, rw, nd, w = self.batch
        self.loss_c = self.native.phase_critic(st, ac, ns, rw, w)

    def _phase2(self):
        self.native.phase_actor(self.batch[0], self.batch[5])

    def _phase3(self):
        self.native.phase_targets()

    def _learn_eager(self):
        self._phase1()
        self.native.allreduce("critic")
        self._phase2()
        self.native.allreduce("actor")
        self._phase3()

    def capture(self, warmup_steps=3):
        """Run `warmup_steps` eager steps (allocator / autotune warm-up, as torch.cuda.graphs requires) and capture."""
        eng, sim = self.eng, self.sim
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
        self.g_pre = torch.cuda.CUDAGraph()
        # thread_local: other threads (the RCCL watchdog of torch.distributed) keep issuing HIP calls during a capture
        mode = dict(capture_error_mode="thread_local")
        with torch.cuda.graph(self.g_pre, **mode):
            eng.pre()
        self.g_post = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_post, **mode):
            eng.post()
        phases = [self._phase1, self._phase2, self._phase3]
        groups = [[p] for p in phases] if self.distributed else [phases]
        pool = None
        for grp in groups:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, **mode):
                for p in grp:
                    p()
            pool = g.pool()
            self.g_learn.append(g)
        torch.cuda.synchronize(self.dev)

    def _learn(self):
        if not self.distributed:
            self.g_learn[0].replay()
        else:
            self.g_learn[0].replay()
            self.native.allreduce("critic")
            self.g_learn[1].replay()
            self.native.allreduce("actor")
            self.g_learn[2].replay()
        self.updates += 1

    def step(self):
        """One env-step for every env + one learner update (once the replay holds episodes)."""
        main, side = self.main, self.side
        self.g_pre.replay()
        learn = self.steps >= self.learn_after
        if learn and self.overlap:
            self.acted.record(main)
        self.sim.step(self.eng.action_t)
        if learn:
            if self.overlap:
                side.wait_event(self.acted)        # weights are free once this step's actor forward is done
                with torch.cuda.stream(side):
                    self._learn()
                main.wait_stream(side)             # the update read the replay before this step's writes
            else:
                self._learn()
        self.g_post.replay()
        self.steps += 1
        return self.eng.reward_out, self.eng.done_out
