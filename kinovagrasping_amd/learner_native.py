"""DDPGfD update without autograd (DDPGfD.train_batch, DDPGfD.py:219-367 of the reference), in two forms:

* **LDS-free MFMA kernels** (widths 256-256 / 128-128 / 64-64; `lds_free`): every forward, data-gradient and
  weight-gradient pass is a hand-written fp32-MFMA launch of libkinova_sim.so (csrc/ks_mlp.hip, mlp.py) that keeps its
  tiles in registers - one wave per 16 batch rows, <= 168 registers per lane, no LDS.  The simulator's stepping kernel
  holds all of every CU's LDS and 344 of the 512 registers per SIMD lane, so these waves are resident BESIDE it: the
  whole update runs in the stepping kernel's shadow on the matrix pipes and issue slots it leaves idle
  (pipeline.GraphedTrainer launches it on a second stream), instead of waiting for its workgroups to retire as the
  library GEMMs (which need LDS) must.  ~35 launches per update.
* **library GEMMs** (any width, e.g. the reference's 400-300): explicit forward / backward GEMMs (PyTorch -> hipBLASLt)
  with the elementwise steps between them as single kernels (kr_relu_backward, ...), weight gradients written by the
  GEMMs straight into one flat gradient buffer per network; the forward-only target networks and the critic forward
  still go through the fused LDS kernel (kr_mlp3_forward) when the width has an instantiation.  ~75 launches.

Common to both: targets + critic loss gradient, Adam and the soft target update are one kernel each (kr_critic_grad,
kr_adam_step, kr_soft_update); with world_size > 1 the flat gradient buffers are all-reduced directly (no pack /
unpack).  Same arithmetic as the reference (critic loss L1 + 0.5 LN on masked row means, actor loss -mean Q(s, pi(s))
over all n-step rows, Adam lr 1e-4 / default lr + weight_decay 1e-4, soft target update every 10th call);
tests/test_gpu_parity.py checks both forms against the autograd implementation (ddpgfd.DDPGfD.train_on_batch).  The
three `phase_*` methods mirror DDPGfD.phase_* so that pipeline.GraphedTrainer can capture them and run the gradient
exchange between them.
"""
from __future__ import annotations

import ctypes

import torch

from . import mlp as _mlp
from . import sim as _sim


class _Net:
    """views of one MLP's flat parameter buffer + a flat gradient buffer of the same layout + Adam state"""

    def __init__(self, module, flat):
        self.flat = flat
        self.grad = torch.zeros_like(flat)
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.W, self.b, self.gW, self.gb = [], [], [], []
        off = 0
        for name in ("l1", "l2", "l3"):
            lin = getattr(module, name)
            for p, dst_p, dst_g in ((lin.weight, self.W, self.gW), (lin.bias, self.b, self.gb)):
                n = p.numel()
                assert p.data.data_ptr() == flat[off:off + n].data_ptr(), "parameters must be views of the flat buffer, in order"
                dst_p.append(p.data)
                dst_g.append(self.grad[off:off + n].view_as(p))
                off += n
        assert off == flat.numel()


class NativeDDPGfDUpdate:
    def __init__(self, policy):
        assert policy.device.type == "cuda", "the learner glue kernels are GPU only"
        self.p = policy
        self.lib = _sim.load_library()
        fp = policy._flat_params
        self.critic, self.actor = _Net(policy.critic, fp["critic"]), _Net(policy.actor, fp["actor"])
        self.critic_t, self.actor_t = _Net(policy.critic_target, fp["critic_target"]), _Net(policy.actor_target, fp["actor_target"])
        self.it = torch.zeros(1, dtype=torch.long, device=policy.device)      # updates done (Adam step of both nets)
        # Pipelined form (phase_head): the actor's Adam step + soft target update of update k run at the START of update
        # k + 1.  `it_head` is k while that step is pending and 0 otherwise, so the head is a no-op when nothing is pending
        # (before the first update, or after finish_pending() has applied it eagerly, e.g. ahead of a checkpoint).
        self.it_head = torch.zeros(1, dtype=torch.long, device=policy.device)
        self.exchange = None             # exchange.PeerExchange when the ranks can map each other's memory (pipeline.GraphedTrainer)
        self.local_gradients = False     # pipeline.AsyncTrainer, replica_sync "average-per-launch": no per-update exchange, replicas averaged per launch
        self.pipelined = False        # set by pipeline.GraphedTrainer: the body marks its actor step pending for the next head
        policy._native = self                                                  # DDPGfD.save / load keep the Adam state in sync
        self.import_optimizer_state()
        self.losses = torch.zeros(3, device=policy.device)                   # critic loss, L1, LN of the last update
        ao, co = policy.actor_optimizer.param_groups[0], policy.critic_optimizer.param_groups[0]
        self.hyper_a = (ao["lr"], ao["betas"][0], ao["betas"][1], ao["eps"], ao["weight_decay"])
        self.hyper_c = (co["lr"], co["betas"][0], co["betas"][1], co["eps"], co["weight_decay"])
        # How much of the update runs on the hand-written MFMA kernels of csrc/ks_mlp.hip (mlp.py):
        #  * fused_targets: the forward-only target networks (3200 rows) and the critic's forward (1600 rows, activations
        #    kept for the backward pass) as ONE launch each instead of three library GEMMs + glue;
        #  * shadow: ... in their LDS-free form, whose waves are resident beside the stepping kernel (which holds every
        #    CU's LDS) instead of waiting for its workgroups to retire;
        #  * lds_free: the backward passes too (mlp.mlp3_backward, mlp.weight_grad) - then no launch of the update needs
        #    LDS and the whole learner runs in the simulator's shadow.  Widths 256-256 / 128-128 / 64-64.
        # Otherwise (e.g. the reference's 400-300) the passes go through the library GEMMs (hipBLASLt).
        import os
        tl = [list(zip(net.W, net.b)) for net in (self.actor_t, self.critic_t)]
        ins = [net.W[0].shape[1] for net in (self.actor_t, self.critic_t)]
        mult = lambda k: all(w.shape[0] % k == 0 for w in self.critic.W[:2] + self.actor.W[:2])
        self.fused_targets = mult(4) and all(_mlp.supported(layers, d) for layers, d in zip(tl, ins))
        self.shadow = self.fused_targets and all(_mlp.supported(layers, d, shadow=True) for layers, d in zip(tl, ins))
        self.lds_free = self.shadow and mult(16) and os.environ.get("KS_EXP_LDSFREE", "1") == "1"
        # The 8000-row forwards of the actor phase stay on the library GEMMs unless the update is LDS-free: at that size
        # three large-tile GEMMs beat the 16-row-tile LDS kernel (measured 1.48 vs 1.44 ms per env-step in bench.py).
        self.fuse_critic_fwd, self.fuse_actor_fwd = True, False
        # diagnostics: the actor loss -mean_w Q(s, pi(s)) is not needed by the update (dLoss/dQ is constant); with
        # track_actor_loss it is evaluated from the Q of phase_actor's critic forward and kept in self.actor_loss
        self.track_actor_loss = False
        self.actor_loss = torch.zeros((), device=policy.device)
        # Fork / join inside a phase (round 5, LDS-free path): the launches of an update that do not depend on each other - the target
        # networks' pass beside the critic's forward, the three weight-gradient launches of a network beside its data-gradient pass -
        # go to two side streams that leave from and rejoin the phase's stream (under stream capture: parallel branches of the graph).
        # Every launch of the LDS-free update is one latency-bound wave per 16 rows (87 us whatever the batch, ks_mlp.hip) on a few
        # hundred of the 1024 SIMDs: branches run at the same time instead of one after the other.
        # OFF by default (KS_LEARNER_FORK=1 turns it on).  Measured, round 5, one MI355X: the update replayed ALONE 0.878 -> 0.733 ms
        # (10.4 GFLOP: 7.5 % -> 9.0 % of the fp32 MFMA peak), bit-identical results (tests/test_gpu_learner_state.py) - but beside the
        # persistent rollout kernel the training step went from 1.34 to 2.09 ms per env-step (3.05 -> 1.96 M env-steps/s) and episodes were
        # dropped: three learner launches at a time compete with the rollout's waves for the SIMDs' issue slots, and the update is off
        # the critical path at one update per env-step anyway.  Kept for learner-bound regimes (updates_per_step >= 2 in lock step).
        self.fork = self.lds_free and os.environ.get("KS_LEARNER_FORK", "0") == "1"
        self._sides = [torch.cuda.Stream(policy.device) for _ in range(2)] if self.fork else []

    # -- helpers ------------------------------------------------------------------------------------------------------
    def _branch(self, k):
        """context: side stream k, ordered behind everything enqueued on the current stream so far (a fork); _join() brings it back"""
        cur = torch.cuda.current_stream(self.p.device)
        side = self._sides[k]
        side.wait_stream(cur)
        return torch.cuda.stream(side)

    def _join(self, *ks):
        cur = torch.cuda.current_stream(self.p.device)
        for k in ks:
            cur.wait_stream(self._sides[k])

    def _st(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.p.device).cuda_stream)

    @staticmethod
    def _chk(rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc})")

    @staticmethod
    def _lin_relu(net, k, x):
        return torch._addmm_activation(net.b[k], x, net.W[k].t())

    def _actor_forward(self, net, x):
        h1 = self._lin_relu(net, 0, x)
        h2 = self._lin_relu(net, 1, h1)
        a = torch.addmm(net.b[2], h2, net.W[2].t()).sigmoid_().mul_(self.p.max_action)
        return h1, h2, a

    def _relu_bwd(self, act, grad):
        self._chk(self.lib.kr_relu_backward(grad.numel(), _sim._ptr(act), _sim._ptr(grad), self._st()), "kr_relu_backward")

    def _adam(self, net, hyper, step=None):
        lr, b1, b2, eps, wd = hyper
        P = _sim._ptr
        self._chk(self.lib.kr_adam_step(net.flat.numel(), P(net.flat), P(net.grad), P(net.exp_avg), P(net.exp_avg_sq),
                                        P(self.it if step is None else step), lr, b1, b2, eps, wd, self._st()), "kr_adam_step")

    # -- optimizer state <-> torch.optim.Adam (the reference's 4-file checkpoint keeps it: DDPGfD.py:371-382) ------------
    def _optim_pairs(self):
        return ((self.actor, self.p.actor_optimizer), (self.critic, self.p.critic_optimizer))

    def export_optimizer_state(self):
        """Make policy.{actor,critic}_optimizer.state describe the native Adam state: exp_avg / exp_avg_sq are VIEWS of the
        flat moment buffers (they stay current), `step` is the number of updates applied.  Applies a pending pipelined
        actor step first, so that both networks are at the same update."""
        torch.cuda.synchronize(self.p.device)          # a pipelined update may still be running on the learner's stream
        self.finish_pending()
        step = float(self.it.item())
        for net, opt in self._optim_pairs():
            off = 0
            for p in opt.param_groups[0]["params"]:
                n = p.numel()
                cap = opt.param_groups[0].get("capturable", False)
                opt.state[p] = {"step": torch.tensor(step, dtype=torch.float32, device=p.device if cap else "cpu"),
                                "exp_avg": net.exp_avg[off:off + n].view_as(p), "exp_avg_sq": net.exp_avg_sq[off:off + n].view_as(p)}
                off += n

    def import_optimizer_state(self):
        """Adopt the state the torch optimizers hold (after DDPGfD.load, or after autograd updates): moments into the flat
        buffers, the update counter from `step`.  No state = a fresh optimizer (zeros)."""
        steps = []
        for net, opt in self._optim_pairs():
            off = 0
            for p in opt.param_groups[0]["params"]:
                n = p.numel()
                stt = opt.state.get(p, None)
                if stt:
                    if stt["exp_avg"].data_ptr() != net.exp_avg[off:off + n].data_ptr():
                        net.exp_avg[off:off + n].copy_(stt["exp_avg"].reshape(-1))
                        net.exp_avg_sq[off:off + n].copy_(stt["exp_avg_sq"].reshape(-1))
                    steps.append(int(float(stt["step"])))
                off += n
        if steps:
            assert min(steps) == max(steps), "actor and critic optimizers are at different steps"
            self.it.fill_(steps[0])
            self.it_head.zero_()
            self.p.total_it = steps[0]

    @torch.no_grad()
    def finish_pending(self):
        """apply the pipelined actor step that is still waiting for the next update's head (no-op otherwise)"""
        self.phase_head()
        self.it_head.zero_()

    def _weight_grads(self, net, x, h1, h2, dz3):
        """dz3: gradient at the last layer's pre-activation; fills net.grad, returns nothing"""
        torch.mm(dz3.t(), h2, out=net.gW[2])
        torch.sum(dz3, 0, out=net.gb[2])
        dh2 = torch.mm(dz3, net.W[2])
        self._relu_bwd(h2, dh2)
        torch.mm(dh2.t(), h1, out=net.gW[1])
        torch.sum(dh2, 0, out=net.gb[1])
        dh1 = torch.mm(dh2, net.W[1])
        self._relu_bwd(h1, dh1)
        torch.mm(dh1.t(), x, out=net.gW[0])
        torch.sum(dh1, 0, out=net.gb[0])

    # -- the three phases ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def phase_critic(self, state, action, next_state, reward, weight=None, next_ends=None):
        """targets, critic forward, loss gradient, critic weight gradients -> self.critic.grad"""
        pol, P = self.p, _sim._ptr
        R = reward.shape[0]
        if weight is None:
            weight = torch.ones(R, device=reward.device)
        # one launch: update counter, sum of the row weights, dLoss/dQ of the actor loss (and, pipelined, the pending mark)
        self.weight, self.wsum = weight, torch.empty(1, device=reward.device)
        self.dq_actor = torch.empty(R * pol.n, 1, device=reward.device)
        self._chk(self.lib.kr_update_prologue(R, pol.n, P(weight), P(self.wsum), P(self.dq_actor), P(self.it), P(self.it_head),
                                              int(self.pipelined), self._st()), "kr_update_prologue")
        # both target evaluations (1-step: next_state[:, 0], n-step: next_state[:, -1]) in one pass of the target nets
        # (next_ends: the sampler's own [2R, S] block of exactly these rows, kr_sample_windows_draw)
        nx = torch.cat([next_state[:, 0], next_state[:, -1]], 0) if next_ends is None else next_ends
        if self.fork:
            # branch 0: the two target networks (3200 rows each); the critic's own forward runs beside them on the phase's stream
            c = self.critic
            cl = list(zip(c.W, c.b))
            s0, a0 = state[:, 0], action[:, 0]
            reward = reward.contiguous()
            with self._branch(0):
                ta = _mlp.mlp3_forward(list(zip(self.actor_t.W, self.actor_t.b)), nx, act=_mlp.ACT_SIGMOID, scale=pol.max_action, shadow=True)
                tq = _mlp.mlp3_forward(list(zip(self.critic_t.W, self.critic_t.b)), nx, ta, act=_mlp.ACT_NONE, shadow=True)
            h1, h2 = s0.new_empty(R, c.W[0].shape[0]), s0.new_empty(R, c.W[1].shape[0])
            q = _mlp.mlp3_forward(cl, s0, a0, act=_mlp.ACT_NONE, h1_out=h1, h2_out=h2, shadow=True)
            dq = torch.empty_like(q)
            self._join(0)
            self._chk(self.lib.kr_critic_grad(R, pol.n, P(q), P(tq), P(tq[R:]), P(reward), P(weight), P(self.wsum), pol.discount, P(dq), P(self.losses),
                                              self._st()), "kr_critic_grad")
            with self._branch(0):                              # the last layer's weight gradient needs dq only: beside the data-gradient pass
                _mlp.weight_grad(dq, h2, None, c.gW[2], c.gb[2])
            dz2, dz1, _ = _mlp.mlp3_backward(cl, dq, h1, h2)
            with self._branch(1):
                _mlp.weight_grad(dz2, h1, None, c.gW[1], c.gb[1])
            _mlp.weight_grad(dz1, s0, a0, c.gW[0], c.gb[0])
            self._join(0, 1)
            return self.losses[0], self.losses[1], self.losses[2]
        if self.fused_targets:
            # forward-only networks: one fused fp32-MFMA launch each (mlp.mlp3_forward) instead of 3 GEMMs + glue
            ta = _mlp.mlp3_forward(list(zip(self.actor_t.W, self.actor_t.b)), nx, act=_mlp.ACT_SIGMOID, scale=pol.max_action, shadow=self.shadow)
            tq = _mlp.mlp3_forward(list(zip(self.critic_t.W, self.critic_t.b)), nx, ta, act=_mlp.ACT_NONE, shadow=self.shadow)
        else:
            _, _, ta = self._actor_forward(self.actor_t, nx)
            ct = self.critic_t
            tq = torch.addmm(ct.b[2], self._lin_relu(ct, 1, self._lin_relu(ct, 0, torch.cat([nx, ta], 1))), ct.W[2].t())
        c = self.critic
        if self.lds_free:
            # the whole critic step without LDS: forward, loss gradient, data and weight gradients (csrc/ks_mlp.hip)
            cl = list(zip(c.W, c.b))
            s0, a0 = state[:, 0], action[:, 0]
            h1, h2 = s0.new_empty(R, c.W[0].shape[0]), s0.new_empty(R, c.W[1].shape[0])
            q = _mlp.mlp3_forward(cl, s0, a0, act=_mlp.ACT_NONE, h1_out=h1, h2_out=h2, shadow=True)
            dq = torch.empty_like(q)
            reward = reward.contiguous()
            self._chk(self.lib.kr_critic_grad(R, pol.n, P(q), P(tq), P(tq[R:]), P(reward), P(weight), P(self.wsum), pol.discount, P(dq), P(self.losses),
                                              self._st()), "kr_critic_grad")
            dz2, dz1, _ = _mlp.mlp3_backward(cl, dq, h1, h2)
            _mlp.weight_grad(dq, h2, None, c.gW[2], c.gb[2])
            _mlp.weight_grad(dz2, h1, None, c.gW[1], c.gb[1])
            _mlp.weight_grad(dz1, s0, a0, c.gW[0], c.gb[0])
            return self.losses[0], self.losses[1], self.losses[2]
        x0 = torch.cat([state[:, 0], action[:, 0]], 1)
        if self.fused_targets and self.fuse_critic_fwd:
            h1, h2 = x0.new_empty(R, c.W[0].shape[0]), x0.new_empty(R, c.W[1].shape[0])
            q = _mlp.mlp3_forward(list(zip(c.W, c.b)), state[:, 0], action[:, 0], act=_mlp.ACT_NONE, h1_out=h1, h2_out=h2, shadow=self.shadow)
        else:
            h1 = self._lin_relu(c, 0, x0)
            h2 = self._lin_relu(c, 1, h1)
            q = torch.addmm(c.b[2], h2, c.W[2].t())
        dq = torch.empty_like(q)
        reward = reward.contiguous()
        self._chk(self.lib.kr_critic_grad(R, pol.n, P(q), P(tq), P(tq[R:]), P(reward), P(weight), P(self.wsum), pol.discount, P(dq), P(self.losses),
                                          self._st()), "kr_critic_grad")
        self._weight_grads(c, x0, h1, h2, dq)
        return self.losses[0], self.losses[1], self.losses[2]

    @torch.no_grad()
    def phase_actor(self, state, weight=None):
        """critic Adam step, then the actor loss -mean_w Q(s, pi(s)) over all n-step rows -> self.actor.grad"""
        pol, P = self.p, _sim._ptr
        self._adam(self.critic, self.hyper_c)
        n = state.shape[1]
        sa = state.reshape(-1, state.shape[2])
        a_, c = self.actor, self.critic
        if self.lds_free:
            rows = sa.shape[0]
            al, cl = list(zip(a_.W, a_.b)), list(zip(c.W, c.b))
            ha1, ha2 = sa.new_empty(rows, a_.W[0].shape[0]), sa.new_empty(rows, a_.W[1].shape[0])
            a = _mlp.mlp3_forward(al, sa, act=_mlp.ACT_SIGMOID, scale=pol.max_action, h1_out=ha1, h2_out=ha2, shadow=True)
            hc1, hc2 = sa.new_empty(rows, c.W[0].shape[0]), sa.new_empty(rows, c.W[1].shape[0])
            q = _mlp.mlp3_forward(cl, sa, a, act=_mlp.ACT_NONE, h1_out=hc1, h2_out=hc2, shadow=True)       # Q itself is not needed
            if self.track_actor_loss:
                self._set_actor_loss(q, n)
            dq = self.dq_actor
            # dLoss/d(actor pre-activation): through the critic to its action inputs, then through 0.8 * sigmoid
            _, _, dz3 = _mlp.mlp3_backward(cl, dq, hc1, hc2, want_dz=False, dx_cols=(sa.shape[1], a.shape[1]), act_out=a, scale=pol.max_action)
            if self.fork:
                with self._branch(0):
                    _mlp.weight_grad(dz3, ha2, None, a_.gW[2], a_.gb[2])
                dz2, dz1, _ = _mlp.mlp3_backward(al, dz3, ha1, ha2)
                with self._branch(1):
                    _mlp.weight_grad(dz2, ha1, None, a_.gW[1], a_.gb[1])
                _mlp.weight_grad(dz1, sa, None, a_.gW[0], a_.gb[0])
                self._join(0, 1)
                return None
            dz2, dz1, _ = _mlp.mlp3_backward(al, dz3, ha1, ha2)
            _mlp.weight_grad(dz3, ha2, None, a_.gW[2], a_.gb[2])
            _mlp.weight_grad(dz2, ha1, None, a_.gW[1], a_.gb[1])
            _mlp.weight_grad(dz1, sa, None, a_.gW[0], a_.gb[0])
            return None
        if self.fused_targets and self.fuse_actor_fwd:
            rows = sa.shape[0]
            ha1, ha2 = sa.new_empty(rows, a_.W[0].shape[0]), sa.new_empty(rows, a_.W[1].shape[0])
            a = _mlp.mlp3_forward(list(zip(a_.W, a_.b)), sa, act=_mlp.ACT_SIGMOID, scale=pol.max_action, h1_out=ha1, h2_out=ha2)
            hc1, hc2 = sa.new_empty(rows, c.W[0].shape[0]), sa.new_empty(rows, c.W[1].shape[0])
            _mlp.mlp3_forward(list(zip(c.W, c.b)), sa, a, act=_mlp.ACT_NONE, h1_out=hc1, h2_out=hc2)      # Q itself is not needed: dLoss/dQ is constant
        else:
            ha1, ha2, a = self._actor_forward(a_, sa)
            hc1 = self._lin_relu(c, 0, torch.cat([sa, a], 1))
            hc2 = self._lin_relu(c, 1, hc1)
        if self.track_actor_loss:
            self._set_actor_loss(torch.addmm(c.b[2], hc2, c.W[2].t()), n)
        # d(-sum_r w_r sum_k Q_rk / (sum(w) n)) / dQ_rk  (kr_update_prologue)
        dq = self.dq_actor
        dh2 = torch.mm(dq, c.W[2])
        self._relu_bwd(hc2, dh2)
        dh1 = torch.mm(dh2, c.W[1])
        self._relu_bwd(hc1, dh1)
        da = torch.mm(dh1, c.W[0][:, sa.shape[1]:])                       # only the action columns of the critic's first layer
        self._chk(self.lib.kr_sigmoid_scale_backward(da.numel(), P(a), pol.max_action, P(da), self._st()), "kr_sigmoid_scale_backward")
        self._weight_grads(a_, sa, ha1, ha2, da)
        return None

    def _set_actor_loss(self, q, n):
        """-sum_r w_r sum_k Q_rk / (sum(w) n)  (DDPGfD.py:341: -critic(state, actor(state)).mean())"""
        self.actor_loss = -(q.view(-1, n).sum(1) * self.weight).sum() / (self.wsum[0] * float(n))

    @torch.no_grad()
    def phase_head(self):
        """Pipelined form of phase_targets: applied at the START of the next update (pipeline.GraphedTrainer), so that
        the actor's weights change at one known, early point of every update instead of at its end.  Gated on the device
        counter `it_head` (set by mark_pending at the end of an update's body, cleared here): a no-op when no actor step is
        pending."""
        pol, P = self.p, _sim._ptr
        self._adam(self.actor, self.hyper_a, step=self.it_head)
        for net, tgt in ((self.critic, self.critic_t), (self.actor, self.actor_t)):
            self._chk(self.lib.kr_soft_update(net.flat.numel(), P(net.flat), P(tgt.flat), pol.tau, P(self.it_head), pol.network_repl_freq, self._st()),
                      "kr_soft_update")
        # (no clearing here: in the pipelined form every body re-marks it_head and a head always runs between two bodies;
        # finish_pending, the eager caller, clears it)

    @torch.no_grad()
    def phase_targets(self):
        """actor Adam step + soft target update on every network_repl_freq-th update"""
        pol, P = self.p, _sim._ptr
        self._adam(self.actor, self.hyper_a)
        pol.total_it += 1
        for net, tgt in ((self.critic, self.critic_t), (self.actor, self.actor_t)):
            self._chk(self.lib.kr_soft_update(net.flat.numel(), P(net.flat), P(tgt.flat), pol.tau, P(self.it), pol.network_repl_freq, self._st()),
                      "kr_soft_update")

    def allreduce(self, net):
        """average the flat gradient buffer of `net` ("critic" / "actor") over the process group (RCCL), in place"""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(self.p.process_group)
        if world == 1 or self.local_gradients:
            return
        g = getattr(self, net).grad
        if self.exchange is not None:
            self.exchange.allreduce_mean(g)        # LDS-free, over peer-mapped memory: runs beside the stepping kernel (exchange.py)
            return
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.p.process_group)
        g.div_(world)

    def train_on_batch(self, state, action, next_state, reward, weight=None):
        losses = self.phase_critic(state, action, next_state, reward, weight)
        self.allreduce("critic")
        self.phase_actor(state, weight)
        self.allreduce("actor")
        self.phase_targets()
        return losses
