"""DDPG-from-Demonstration learner (L1-L3) for the batched simulator.

Mirrors gym-kinova-gripper/DDPGfD.py of the reference: Actor `max_action*sigmoid(l3 relu(l2 relu(l1 s)))`
(DDPGfD.py:15-32), Critic on [state, action] (DDPGfD.py:35-50), `select_action` (71-73) and the working
update `train_batch` (219-367): 1-step + n-step critic targets, lambda_1 = 0.5, Adam (actor lr 1e-4,
critic default lr + weight_decay 1e-4, DDPGfD.py:57,61), soft target update every 10th call
(360-366).  The broken full-episode `train` (SURVEY note N2) is not reproduced.

Layer names (l1, l2, l3) and shapes match the reference so its 4-file checkpoints
(`<prefix>_{actor,critic,actor_optimizer,critic_optimizer}`, DDPGfD.py:371-382) load unchanged.  Hidden
widths are parameters: 400-300 is the reference (parity), 256-256 is what BASELINE.json benchmarks.

MI355X notes: everything stays on the GPU (no per-step host round trip as in DDPGfD.py:71-73); the
GEMMs run on the matrix cores through PyTorch-ROCm (hipBLASLt); with world_size > 1 the gradients of
both networks are averaged with ONE all-reduce over a flat fp32 buffer (RCCL over xGMI).
"""
from __future__ import annotations

import copy

import torch
import torch.nn as nn
import torch.nn.functional as F


class Actor(nn.Module):
    def __init__(self, state_dim, action_dim, max_action, hidden=(400, 300)):
        super().__init__()
        self.l1 = nn.Linear(state_dim, hidden[0])
        nn.init.kaiming_uniform_(self.l1.weight, a=0, mode="fan_in", nonlinearity="leaky_relu")
        self.l2 = nn.Linear(hidden[0], hidden[1])
        nn.init.kaiming_uniform_(self.l2.weight, a=0, mode="fan_in", nonlinearity="leaky_relu")
        self.l3 = nn.Linear(hidden[1], action_dim)
        nn.init.kaiming_uniform_(self.l3.weight, a=0, mode="fan_in", nonlinearity="leaky_relu")
        self.max_action = max_action

    def forward(self, state):
        a = F.relu(self.l1(state))
        a = F.relu(self.l2(a))
        return self.max_action * torch.sigmoid(self.l3(a))


class Critic(nn.Module):
    def __init__(self, state_dim, action_dim, hidden=(400, 300)):
        super().__init__()
        self.l1 = nn.Linear(state_dim + action_dim, hidden[0])
        nn.init.kaiming_uniform_(self.l1.weight, a=0, mode="fan_in", nonlinearity="leaky_relu")
        self.l2 = nn.Linear(hidden[0], hidden[1])
        nn.init.kaiming_uniform_(self.l2.weight, a=0, mode="fan_in", nonlinearity="leaky_relu")
        self.l3 = nn.Linear(hidden[1], 1)
        nn.init.kaiming_uniform_(self.l3.weight, a=0, mode="fan_in", nonlinearity="leaky_relu")

    def forward(self, state, action):
        q = F.relu(self.l1(torch.cat([state, action], -1)))
        q = F.relu(self.l2(q))
        return self.l3(q)


def _flatten_parameters(module: nn.Module) -> torch.Tensor:
    """Re-home the parameters of `module` as views into ONE flat fp32 buffer (returned): the soft target update is
    then a single elementwise expression per network instead of one per tensor."""
    params = list(module.parameters())
    flat = torch.cat([p.data.reshape(-1) for p in params])
    off = 0
    for p in params:
        n = p.numel()
        p.data = flat[off:off + n].view_as(p)
        off += n
    return flat


class DDPGfD:
    def __init__(self, state_dim=82, action_dim=4, max_action=0.8, n=5, discount=0.995, tau=0.0005, batch_size=64,
                 hidden=(400, 300), device="cpu", process_group=None, capturable=False):
        self.device = torch.device(device)
        self.actor = Actor(state_dim, action_dim, max_action, hidden).to(self.device)
        self.actor_target = copy.deepcopy(self.actor)
        self._flat_params = {}
        # capturable: optimizer state and the update counter live on the device, so that a whole update can be
        # captured in a HIP graph (pipeline.GraphedTrainer); the arithmetic is the same
        self.capturable = bool(capturable)
        self.actor_optimizer = torch.optim.Adam(self.actor.parameters(), lr=1e-4, capturable=self.capturable)
        self.critic = Critic(state_dim, action_dim, hidden).to(self.device)
        self.critic_target = copy.deepcopy(self.critic)
        self.critic_optimizer = torch.optim.Adam(self.critic.parameters(), weight_decay=1e-4, capturable=self.capturable)
        self._it_dev = torch.zeros((), dtype=torch.long, device=self.device)
        for name in ("actor", "actor_target", "critic", "critic_target"):
            self._flat_params[name] = _flatten_parameters(getattr(self, name))
        self.discount, self.tau, self.n = discount, tau, n
        self._disc = torch.tensor([discount ** i for i in range(n)], dtype=torch.float32, device=self.device)
        self.network_repl_freq = 10
        self.total_it = 0
        self.batch_size = batch_size
        self.max_action = max_action
        self.process_group = process_group
        self._flat = None
        self._native = None          # learner_native.NativeDDPGfDUpdate attached to this policy (keeps its own flat Adam state)

    # -- inference ------------------------------------------------------------------------------
    @torch.no_grad()
    def select_action(self, state):
        """state: [..., 82] tensor/array -> actions [..., 4] (stays on the device for tensors)"""
        if not torch.is_tensor(state):
            s = torch.as_tensor(state, dtype=torch.float32, device=self.device).reshape(1, -1)
            return self.actor(s).cpu().numpy().flatten()
        return self.actor(state)

    # -- gradient exchange: one all-reduce over a flat buffer holding both networks ---
    def _allreduce_grads(self, params):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(self.process_group)
        if world == 1:
            return
        grads = [p.grad for p in params]
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.process_group)
        flat.div_(world)
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()

    # -- update ---------------------------------------------------------------------------------
    def train_on_batch(self, state, action, next_state, reward, weight=None):
        """One DDPGfD update on already-sampled n-step windows: state/next_state [R, n, 82],
        action [R, n, 4], reward [R, n].  `weight` [R] (optional, 0/1) masks padding rows so that
        fixed-shape device batches keep the reference's per-row means.  Returns the four losses
        (actor, critic, critic_L1, critic_LN) as 0-d tensors (no host sync).

        The update is three phases with a gradient exchange after the first two (`phase_critic`,
        `phase_actor`, `phase_targets`); pipeline.GraphedTrainer captures the phases in HIP graphs and runs the
        exchanges between them."""
        losses_c = self.phase_critic(state, action, next_state, reward, weight)
        self._allreduce_grads(list(self.critic.parameters()))
        actor_loss = self.phase_actor(state, weight)
        self._allreduce_grads(list(self.actor.parameters()))
        self.phase_targets()
        return (actor_loss,) + losses_c

    @staticmethod
    def _mean(x, per_row, weight):
        if weight is None:
            return x.mean()
        w = weight.view(-1, *([1] * (x.dim() - 1)))
        return (x * w).sum() / (w.sum().clamp_min(1.0) * per_row)      # 0 / 1 weights: exact unless the batch is all padding

    def phase_critic(self, state, action, next_state, reward, weight=None):
        """targets + critic loss + backward (DDPGfD.py:256-330).  Returns (critic, L1, LN) losses."""
        reward = reward.unsqueeze(-1)
        with torch.no_grad():
            # both target evaluations (1-step: next_state[:, 0], n-step: next_state[:, -1]) in one pass of the target nets
            R = reward.shape[0]
            nx = torch.cat([next_state[:, 0], next_state[:, -1]], 0)
            tq = self.critic_target(nx, self.actor_target(nx))
            target_Q = reward[:, 0] + self.discount * tq[:R]
            n_step_return = (reward.squeeze(-1) * self._disc).sum(1)
            target_QN = (n_step_return + (self.discount ** self.n) * tq[R:].squeeze(-1)).unsqueeze(-1)
        current_Q = self.critic(state[:, 0], action[:, 0])
        critic_L1 = self._mean((current_Q - target_Q) ** 2, 1, weight)
        critic_LN = self._mean((current_Q - target_QN) ** 2, 1, weight)
        critic_loss = critic_L1 + 0.5 * critic_LN
        self.critic_optimizer.zero_grad()
        critic_loss.backward()
        return critic_loss.detach(), critic_L1.detach(), critic_LN.detach()

    def phase_actor(self, state, weight=None):
        """critic step, then actor loss + backward (DDPGfD.py:331-352)"""
        self.critic_optimizer.step()
        # the critic is only a differentiable function here: no weight gradients for it (the reference computes and
        # discards them)
        for p in self.critic.parameters():
            p.requires_grad_(False)
        actor_loss = -self._mean(self.critic(state, self.actor(state)), state.shape[1], weight)
        self.actor_optimizer.zero_grad()
        actor_loss.backward()
        for p in self.critic.parameters():
            p.requires_grad_(True)
        return actor_loss.detach()

    def phase_targets(self):
        """actor step + soft target update on every 10th call (DDPGfD.py:353-366)"""
        self.actor_optimizer.step()
        self.total_it += 1
        fp = self._flat_params
        pairs = ((fp["critic"], fp["critic_target"]), (fp["actor"], fp["actor_target"]))
        with torch.no_grad():
            if self.capturable:
                # device-side gate: tau on every network_repl_freq-th call, else 0 (same arithmetic when it fires)
                self._it_dev += 1
                fire = self._it_dev % self.network_repl_freq == 0
                for p, tp in pairs:
                    tp.copy_(torch.where(fire, self.tau * p + (1 - self.tau) * tp, tp))
            elif self.total_it % self.network_repl_freq == 0:
                for p, tp in pairs:
                    tp.copy_(self.tau * p + (1 - self.tau) * tp)

    def train_batch(self, episode_step, expert_replay_buffer, replay_buffer, num_trajectories=5, prob=0.3):
        """Reference signature (DDPGfD.py:219): samples agent (1-prob) / expert (prob) episodes from
        buffers exposing sample_batch_nstep(batch_size) and runs one update.  Returns floats."""
        if replay_buffer is not None and expert_replay_buffer is None:
            batch = replay_buffer.sample_batch_nstep(self.batch_size)
        elif replay_buffer is None and expert_replay_buffer is not None:
            batch = expert_replay_buffer.sample_batch_nstep(self.batch_size)
        else:
            agent_bs = int(self.batch_size * (1 - prob))
            ag = replay_buffer.sample_batch_nstep(agent_bs)
            ex = expert_replay_buffer.sample_batch_nstep(self.batch_size - agent_bs)
            batch = tuple(torch.cat((a, e), 0) for a, e in zip(ag, ex))
        state, action, next_state, reward = (t.to(self.device) for t in batch[:4])
        weight = batch[5].to(self.device) if len(batch) > 5 else None
        return tuple(x.item() for x in self.train_on_batch(state, action, next_state, reward, weight))

    # -- checkpoints: the reference's four files (DDPGfD.py:371-382) ---------------------------------
    def save(self, filename):
        if self._native is not None:
            self._native.export_optimizer_state()     # the native learner's Adam moments / step -> the torch optimizers' state
        torch.save(self.critic.state_dict(), filename + "_critic")
        torch.save(self.critic_optimizer.state_dict(), filename + "_critic_optimizer")
        torch.save(self.actor.state_dict(), filename + "_actor")
        torch.save(self.actor_optimizer.state_dict(), filename + "_actor_optimizer")

    def load(self, filename, weights_only=True, sync_targets=False):
        """Like the reference, loading leaves the target networks untouched (DDPGfD.py:378-382);
        sync_targets=True copies the loaded weights into them."""
        import os
        self.critic.load_state_dict(torch.load(filename + "_critic", map_location=self.device, weights_only=weights_only))
        self.actor.load_state_dict(torch.load(filename + "_actor", map_location=self.device, weights_only=weights_only))
        if os.path.exists(filename + "_critic_optimizer"):
            self.critic_optimizer.load_state_dict(torch.load(filename + "_critic_optimizer", map_location=self.device, weights_only=weights_only))
        if os.path.exists(filename + "_actor_optimizer"):
            self.actor_optimizer.load_state_dict(torch.load(filename + "_actor_optimizer", map_location=self.device, weights_only=weights_only))
        if sync_targets:
            self.critic_target.load_state_dict(self.critic.state_dict())
            self.actor_target.load_state_dict(self.actor.state_dict())
        if self._native is not None:
            self._native.import_optimizer_state()     # ... and back: the native learner continues from the loaded Adam state
