"""Episode replay for n-step DDPGfD (L4): restatement of ReplayBuffer_Queue (gym-kinova-gripper/utils.py:9-343)
as (a) a host sampler that consumes the same np.random stream as the reference (golden-vector parity)
and (b) a device-resident, fixed-shape episode ring for the batched rollout.

Reference semantics kept (utils.py:240-306): episodes drawn with np.random.randint(replay_ep_num - 1)
(the newest episode is never sampled); for an episode of length L, ceiling = L - n windows:
ceiling - 1 uniform random starts in [0, ceiling) plus the final window starting at `ceiling`.
"""
from __future__ import annotations

import numpy as np
import torch


def sample_windows_host(ep_lens, n_steps: int, batch_size: int, rng=np.random):
    """Exactly the draws of ReplayBuffer_Queue.sample_batch_nstep: returns [(episode, start), ...]."""
    out = []
    if batch_size < 1:
        return out
    episode_idx = rng.randint(len(ep_lens) - 1, size=batch_size)
    for idx in episode_idx:
        ceiling = int(ep_lens[idx]) - n_steps
        for _ in range(ceiling - 1):
            out.append((int(idx), int(rng.randint(ceiling))))
        out.append((int(idx), ceiling))
    return out


# ---- the reference's on-disk replay bundle (utils.py:345-390): a directory of .npy files, each the pickled
# nested list the buffer holds in memory - state / action / next_state / reward / not_done: one list per episode
# (+ the empty list of the episode being filled), episodes: [first, last] timestep counters per episode,
# episodes_info: [max_episode, size, episodes_count, replay_ep_num], orientation_indexes.
_BUNDLE_FIELDS = ("state", "action", "next_state", "reward", "not_done")


def _object_rows(rows):
    out = np.empty(len(rows), dtype=object)
    for i, r in enumerate(rows):
        out[i] = r
    return out


def save_reference_bundle(dirpath, episodes, max_episode=10000, orientation_indexes=None):
    """Write episodes (list of dicts with the five _BUNDLE_FIELDS, each [L, ...] array) in the layout
    ReplayBuffer_Queue.save_replay_buffer produces, readable by store_saved_data_into_replay (utils.py:367-400)."""
    from pathlib import Path
    d = Path(dirpath)
    d.mkdir(parents=True, exist_ok=True)
    for f in _BUNDLE_FIELDS:
        rows = []
        for ep in episodes:
            a = np.asarray(ep[f])
            rows.append([np.asarray(x, dtype=np.float64) if a.ndim > 1 else float(x) for x in a])
        rows.append([])                                         # the open episode
        np.save(d / f, _object_rows(rows), allow_pickle=True)
    lens = [len(np.asarray(ep["reward"])) for ep in episodes]
    np.save(d / "episodes", _object_rows([[0, n] for n in lens] + [[]]), allow_pickle=True)
    np.save(d / "episodes_info", np.array([max_episode, sum(lens), len(lens), len(lens)]))
    np.save(d / "orientation_indexes", _object_rows(list(orientation_indexes) if orientation_indexes is not None else []), allow_pickle=True)


def load_reference_bundle(dirpath):
    """Read a bundle written by the reference (or by save_reference_bundle): list of episode dicts of float32 arrays
    (empty trailing episodes dropped) + the episodes_info vector."""
    from pathlib import Path
    d = Path(dirpath)
    cols = {f: np.load(d / (f + ".npy"), allow_pickle=True) for f in _BUNDLE_FIELDS}
    n = len(cols["reward"])
    episodes = []
    for i in range(n):
        if len(cols["reward"][i]) == 0:
            continue
        episodes.append({f: np.asarray([np.asarray(x, dtype=np.float32) for x in cols[f][i]], dtype=np.float32) for f in _BUNDLE_FIELDS})
    info = np.load(d / "episodes_info.npy", allow_pickle=True)
    return episodes, np.asarray(info, dtype=np.int64)


class HostEpisodeReplay:
    """Flat-array host replay with the reference's sampler; used for expert data and parity tests."""

    def __init__(self, state_dim=82, action_dim=4, n_steps=5):
        self.n_steps = n_steps
        self.lens, self.offsets = [], []
        self.state, self.action, self.next_state, self.reward, self.not_done = [], [], [], [], []

    def add_episode_arrays(self, state, action, next_state, reward, not_done):
        self.offsets.append(sum(self.lens))
        self.lens.append(len(reward))
        for lst, arr in ((self.state, state), (self.action, action), (self.next_state, next_state), (self.reward, reward), (self.not_done, not_done)):
            lst.append(np.asarray(arr, dtype=np.float32))

    @property
    def replay_ep_num(self):
        return len(self.lens)

    def episodes(self):
        return [dict(state=self.state[i], action=self.action[i], next_state=self.next_state[i], reward=self.reward[i], not_done=self.not_done[i])
                for i in range(len(self.lens))]

    def save(self, dirpath, max_episode=10000):
        """the reference's replay bundle (utils.py:345-365)"""
        save_reference_bundle(dirpath, self.episodes(), max_episode)

    def load(self, dirpath):
        """append the episodes of a reference replay bundle (utils.py:367-400)"""
        eps, info = load_reference_bundle(dirpath)
        for ep in eps:
            self.add_episode_arrays(ep["state"], ep["action"], ep["next_state"], ep["reward"], ep["not_done"])
        return info

    def sample_batch_nstep(self, batch_size, num_ts_from_ep=5, rng=np.random):
        wins = sample_windows_host(self.lens, self.n_steps, batch_size, rng)
        n = self.n_steps
        pick = lambda lst: torch.from_numpy(np.stack([lst[e][s:s + n] for e, s in wins])) if wins else torch.zeros(0)
        return pick(self.state), pick(self.action), pick(self.next_state), pick(self.reward), pick(self.not_done)


class DeviceEpisodeReplay:
    """Fixed-shape episode ring on the GPU: [capacity, horizon, ...] plus per-episode lengths.
    Envs append to their own open episode; a finished episode is committed to the ring (FIFO).

    Every method is a fixed sequence of fixed-shape device ops - no host synchronisation, no data-dependent
    shapes - so a whole rollout step can be captured in a HIP graph.  Masked rows are handled with
    torch.where; episodes that are not committed are copied to a trash row behind the ring."""

    def __init__(self, n_envs, capacity, horizon=30, state_dim=82, action_dim=4, n_steps=5, device="cuda"):
        self.n_envs, self.capacity, self.horizon, self.n_steps = n_envs, capacity, horizon, n_steps
        self.device = torch.device(device)
        z = lambda *s: torch.zeros(*s, device=self.device)
        rows = capacity + 1                                # row `capacity` is the trash row
        self.ep_state, self.ep_next = z(rows, horizon, state_dim), z(rows, horizon, state_dim)
        self.ep_action, self.ep_reward, self.ep_not_done = z(rows, horizon, action_dim), z(rows, horizon), z(rows, horizon)
        self.ep_len = torch.zeros(rows, dtype=torch.long, device=self.device)
        self._count = torch.zeros((), dtype=torch.long, device=self.device)   # committed episodes
        self._head = torch.zeros((), dtype=torch.long, device=self.device)    # next ring slot
        self.cur_state, self.cur_next = z(n_envs, horizon, state_dim), z(n_envs, horizon, state_dim)
        self.cur_action, self.cur_reward, self.cur_not_done = z(n_envs, horizon, action_dim), z(n_envs, horizon), z(n_envs, horizon)
        self.cur_len = torch.zeros(n_envs, dtype=torch.long, device=self.device)
        self._env_ar = torch.arange(n_envs, device=self.device)
        self._all = torch.ones(n_envs, dtype=torch.bool, device=self.device)
        self._row = torch.arange(horizon - n_steps, device=self.device).unsqueeze(0)
        self._win = torch.arange(n_steps, device=self.device)
        # on a GPU the bookkeeping runs as the kr_* kernels of libkinova_sim.so (include/kinova_rollout.h), one launch
        # per method instead of a dozen torch ops; the torch code below is the same arithmetic (and their checker)
        self.native = self.device.type == "cuda"
        if self.native:
            from . import sim as _sim
            self._lib, self._ptr = _sim.load_library(), _sim._ptr
            self._rank = torch.zeros(n_envs, dtype=torch.long, device=self.device)
            self._total = torch.zeros(1, dtype=torch.long, device=self.device)

    def _stream(self):
        import ctypes
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc})")

    @property
    def count(self):
        """committed episodes (host read: synchronises)"""
        return int(self._count)

    @property
    def head(self):
        return int(self._head)

    def add(self, state, action, next_state, reward, done, store_mask=None):
        """Append one transition per env (where store_mask is True; utils.py:34-64)."""
        m = self._all if store_mask is None else store_mask
        ar, t = self._env_ar, self.cur_len.clamp(max=self.horizon - 1)
        mc = m.unsqueeze(1)
        self.cur_state[ar, t] = torch.where(mc, state, self.cur_state[ar, t])
        self.cur_next[ar, t] = torch.where(mc, next_state, self.cur_next[ar, t])
        self.cur_action[ar, t] = torch.where(mc, action, self.cur_action[ar, t])
        self.cur_reward[ar, t] = torch.where(m, reward, self.cur_reward[ar, t])
        self.cur_not_done[ar, t] = torch.where(m, 1.0 - done.float(), self.cur_not_done[ar, t])
        self.cur_len.copy_(torch.where(m, t + 1, self.cur_len))

    def replace_last(self, env_mask, reward):
        """utils.py:309-343: overwrite the last stored transition of the open episode with the lift outcome."""
        m = env_mask & (self.cur_len > 0)
        ar, last = self._env_ar, (self.cur_len - 1).clamp(min=0)
        self.cur_reward[ar, last] = torch.where(m, reward, self.cur_reward[ar, last])
        self.cur_not_done[ar, last] = torch.where(m, torch.zeros_like(reward), self.cur_not_done[ar, last])

    def end_episodes(self, env_mask):
        """Commit the open episodes of `env_mask` envs in env order; episodes with len - n <= 1 are dropped
        (main_DDPGfD.py:469-471).  Returns the number committed as a 0-d device tensor."""
        keep = env_mask & (self.cur_len - self.n_steps > 1)
        if self.native:
            return self.commit_native(keep, env_mask)
        rank = torch.cumsum(keep.long(), 0) - 1
        slots = torch.where(keep, (self._head + rank) % self.capacity, torch.full_like(rank, self.capacity))
        self.ep_state.index_copy_(0, slots, self.cur_state)
        self.ep_next.index_copy_(0, slots, self.cur_next)
        self.ep_action.index_copy_(0, slots, self.cur_action)
        self.ep_reward.index_copy_(0, slots, self.cur_reward)
        self.ep_not_done.index_copy_(0, slots, self.cur_not_done)
        self.ep_len.index_copy_(0, slots, self.cur_len)
        k = keep.sum()
        self._head.copy_((self._head + k) % self.capacity)
        self._count.copy_((self._count + k).clamp(max=self.capacity))
        self.cur_len.copy_(torch.where(env_mask, torch.zeros_like(self.cur_len), self.cur_len))
        return k

    def host_episodes(self):
        """committed episodes, oldest first, as numpy dicts (synchronises)"""
        cnt, head = self.count, self.head
        first = (head - cnt) % self.capacity
        out = []
        for k in range(cnt):
            s = (first + k) % self.capacity
            L = int(self.ep_len[s])
            out.append(dict(state=self.ep_state[s, :L].cpu().numpy(), action=self.ep_action[s, :L].cpu().numpy(),
                            next_state=self.ep_next[s, :L].cpu().numpy(), reward=self.ep_reward[s, :L].cpu().numpy(),
                            not_done=self.ep_not_done[s, :L].cpu().numpy()))
        return out

    def save(self, dirpath, max_episode=None):
        """the reference's replay bundle (utils.py:345-365)"""
        save_reference_bundle(dirpath, self.host_episodes(), self.capacity if max_episode is None else max_episode)

    def load(self, dirpath):
        """append the episodes of a reference replay bundle to the ring (episodes longer than the horizon are cut)"""
        eps, info = load_reference_bundle(dirpath)
        for ep in eps:
            L = min(len(ep["reward"]), self.horizon)
            s = int(self._head)
            for name, key in (("ep_state", "state"), ("ep_next", "next_state"), ("ep_action", "action"), ("ep_reward", "reward"), ("ep_not_done", "not_done")):
                getattr(self, name)[s, :L] = torch.as_tensor(ep[key][:L]).to(self.device)
            self.ep_len[s] = L
            self._head.copy_((self._head + 1) % self.capacity)
            self._count.copy_((self._count + 1).clamp(max=self.capacity))
        return info

    def enable_async(self):
        """Open-episode buffers of the free-running rollout kernel (ks_rollout): TWO per env, so that an env can start its next
        episode while the learner's stream has not yet moved the finished one into the ring.  a_* [2, n, H, ...], a_len [2, n],
        a_sel [n] (which one is open), pub_len [2, n] (> 0: a finished episode of that length awaits commit_published)."""
        if hasattr(self, "pub_len"):
            return
        n, H, dev = self.n_envs, self.horizon, self.device
        S, A = self.ep_state.shape[2], self.ep_action.shape[2]
        z = lambda *sh, **k: torch.zeros(*sh, device=dev, **k)
        self.a_state, self.a_next, self.a_action = z(2, n, H, S), z(2, n, H, S), z(2, n, H, A)
        self.a_reward, self.a_not_done = z(2, n, H), z(2, n, H)
        self.a_len, self.pub_len = z(2, n, dtype=torch.long), z(2, n, dtype=torch.long)
        self.a_sel = z(n, dtype=torch.uint8)
        self._keep2 = torch.zeros(2, n, dtype=torch.bool, device=dev)

    def commit_published(self):
        """Move every published episode (pub_len > 0) of both buffers into the ring, buffer 0 first, env order within a buffer,
        and free the buffers (kr_rank_episodes / kr_commit_episodes / kr_advance_ring with the published lengths as cur_len).
        Fixed-shape device ops: capturable; runs on the learner's stream ahead of the window sampling."""
        L, P, st = self._lib, self._ptr, self._stream()
        for b in (0, 1):
            keep = self._keep2[b]
            torch.gt(self.pub_len[b], 0, out=keep)
            self._check(L.kr_rank_episodes(self.n_envs, P(keep), P(self._rank), P(self._total), st), "kr_rank_episodes")
            self._check(L.kr_commit_episodes(self.n_envs, self.horizon, self.capacity, P(keep), P(self._rank), P(self._head), P(self.a_state[b]),
                                             P(self.a_next[b]), P(self.a_action[b]), P(self.a_reward[b]), P(self.a_not_done[b]), P(self.pub_len[b]),
                                             P(self.ep_state), P(self.ep_next), P(self.ep_action), P(self.ep_reward), P(self.ep_not_done),
                                             P(self.ep_len), st), "kr_commit_episodes")
            self._check(L.kr_advance_ring(self.n_envs, self.capacity, P(self._total), P(self._head), P(self._count), P(keep), P(self.pub_len[b]), st),
                        "kr_advance_ring")

    def commit_native(self, keep, ended):
        """rank -> commit -> advance with the kr_* kernels; keep / ended: bool [n_envs]"""
        L, P, st = self._lib, self._ptr, self._stream()
        self._check(L.kr_rank_episodes(self.n_envs, P(keep), P(self._rank), P(self._total), st), "kr_rank_episodes")
        self._check(L.kr_commit_episodes(self.n_envs, self.horizon, self.capacity, P(keep), P(self._rank), P(self._head), P(self.cur_state),
                                         P(self.cur_next), P(self.cur_action), P(self.cur_reward), P(self.cur_not_done), P(self.cur_len),
                                         P(self.ep_state), P(self.ep_next), P(self.ep_action), P(self.ep_reward), P(self.ep_not_done),
                                         P(self.ep_len), st), "kr_commit_episodes")
        self._check(L.kr_advance_ring(self.n_envs, self.capacity, P(self._total), P(self._head), P(self._count), P(ended), P(self.cur_len), st),
                    "kr_advance_ring")
        return self._total[0]

    def _ring(self):
        from .sim import KrRing
        P = lambda t: t.data_ptr()
        return KrRing(P(self._count), P(self._head), self.capacity, P(self.ep_len), P(self.ep_state), P(self.ep_next), P(self.ep_action),
                      P(self.ep_reward), P(self.ep_not_done))

    def sample_mixed(self, expert: "DeviceEpisodeReplay", batch_size, prob=0.3, uniforms=None, draw=None, seed=0, generator=None):
        """DDPGfD's batch (DDPGfD.train_batch, DDPGfD.py:232-254): agent_batch_size = int(batch_size * (1 - prob)) episodes from THIS
        ring followed by batch_size - agent_batch_size episodes from `expert`, each sampled with sample_batch_nstep's rule on its own
        ring.  Same return layout as sample_batch_nstep (incl. the 7th tensor when `draw` is given).  One kernel launch on the GPU
        (kr_sample_windows_mixed); the torch path below - the concatenation of two sample_batch_nstep calls - is its checker."""
        n, W = self.n_steps, self.horizon - self.n_steps
        if expert.horizon != self.horizon or expert.n_steps != n:
            raise ValueError("sample_mixed: the expert ring must have the agent ring's horizon and n_steps")
        b_agent = int(batch_size * (1 - prob))
        b_exp = batch_size - b_agent
        if self.native and expert.native:
            import ctypes
            R, dev = batch_size * W, self.device
            S, A = self.ep_state.shape[2], self.ep_action.shape[2]
            out = (torch.empty(R, n, S, device=dev), torch.empty(R, n, A, device=dev), torch.empty(R, n, S, device=dev),
                   torch.empty(R, n, device=dev), torch.empty(R, n, device=dev), torch.empty(R, device=dev))
            P = self._ptr
            ends = torch.empty(2 * R, S, device=dev) if (draw is not None and uniforms is None) else None
            if uniforms is None and draw is None:
                uniforms = torch.rand(batch_size * (W + 1), device=dev, generator=generator)
            u = None if uniforms is None else uniforms.contiguous()
            ra, re = self._ring(), expert._ring()
            self._check(self._lib.kr_sample_windows_mixed(batch_size, b_agent, self.horizon, n, ctypes.byref(ra), ctypes.byref(re),
                                                          P(u), P(u[batch_size:]) if u is not None else None, int(seed) & (2 ** 64 - 1),
                                                          P(draw) if u is None else None, P(out[0]), P(out[1]), P(out[2]), P(out[3]), P(out[4]),
                                                          P(out[5]), P(ends), self._stream()), "kr_sample_windows_mixed")
            return out if ends is None else out + (ends,)
        if uniforms is None:
            uniforms = torch.rand(batch_size * (W + 1), device=self.device, generator=generator)
        ue, us = uniforms[:batch_size], uniforms[batch_size:].view(batch_size, W)
        parts = []
        if b_agent:
            parts.append(self.sample_batch_nstep(b_agent, uniforms=torch.cat([ue[:b_agent], us[:b_agent].reshape(-1)])))
        if b_exp:
            parts.append(expert.sample_batch_nstep(b_exp, uniforms=torch.cat([ue[b_agent:], us[b_agent:].reshape(-1)])))
        return tuple(torch.cat([p[k] for p in parts], 0) for k in range(6))

    def sample_batch_nstep(self, batch_size, generator=None, uniforms=None, draw=None, seed=0):
        """Fixed-shape batch: batch_size episodes x (horizon - n) window rows, padding rows have weight 0.
        Returns state [R,n,S], action [R,n,A], next_state [R,n,S], reward [R,n], not_done [R,n], weight [R].
        `uniforms` (optional, tests): the batch_size + batch_size*W numbers in [0,1) to use instead of torch.rand.
        `draw` (native path): a device int64 counter that differs from call to call (the learner's update count) - the
        uniforms are then drawn inside the kernel (Philox keyed by `seed`; no torch generator, hence none of its state
        launches in a captured graph) and a 7th tensor is returned: next_state[:, 0] and next_state[:, -1] stacked
        [2R, S], the rows the target networks evaluate."""
        n, W = self.n_steps, self.horizon - self.n_steps
        if self.native:
            R, dev = batch_size * W, self.device
            S, A = self.ep_state.shape[2], self.ep_action.shape[2]
            out = (torch.empty(R, n, S, device=dev), torch.empty(R, n, A, device=dev), torch.empty(R, n, S, device=dev),
                   torch.empty(R, n, device=dev), torch.empty(R, n, device=dev), torch.empty(R, device=dev))
            P = self._ptr
            if draw is not None and uniforms is None:
                ends = torch.empty(2 * R, S, device=dev)
                self._check(self._lib.kr_sample_windows_draw(batch_size, self.horizon, n, P(self._count), P(self._head), self.capacity, P(self.ep_len),
                                                             int(seed) & (2 ** 64 - 1), P(draw), P(self.ep_state), P(self.ep_next), P(self.ep_action),
                                                             P(self.ep_reward), P(self.ep_not_done), P(out[0]), P(out[1]), P(out[2]), P(out[3]),
                                                             P(out[4]), P(out[5]), P(ends), self._stream()), "kr_sample_windows_draw")
                return out + (ends,)
            u = torch.rand(batch_size * (W + 1), device=self.device, generator=generator) if uniforms is None else uniforms.contiguous()
            self._check(self._lib.kr_sample_windows(batch_size, self.horizon, n, P(self._count), P(self._head), self.capacity, P(self.ep_len), P(u),
                                                    P(u[batch_size:]),
                                                    P(self.ep_state), P(self.ep_next), P(self.ep_action), P(self.ep_reward),
                                                    P(self.ep_not_done), P(out[0]), P(out[1]), P(out[2]), P(out[3]), P(out[4]), P(out[5]),
                                                    self._stream()), "kr_sample_windows")
            return out
        # the k-th oldest episode, k in [0, count - 1): the newest one (ring slot head - 1) is never sampled (utils.py:259)
        hi = (self._count - 1).clamp(min=1)
        ue = torch.rand(batch_size, device=self.device, generator=generator) if uniforms is None else uniforms[:batch_size]
        k = torch.minimum((ue * hi).long(), hi - 1)
        ep = (self._head - self._count + k) % self.capacity
        ceiling = (self.ep_len[ep] - n).clamp(min=1)       # [B]
        row = self._row                                                      # [1,W]
        u = torch.rand(batch_size, W, device=self.device, generator=generator) if uniforms is None else uniforms[batch_size:].view(batch_size, W)
        start = (u * ceiling.unsqueeze(1)).long().clamp(max=self.horizon - n)
        start = torch.where(row == (ceiling.unsqueeze(1) - 1), ceiling.unsqueeze(1).expand(-1, W), start)
        start = start.clamp(max=self.horizon - n)
        weight = ((row < ceiling.unsqueeze(1)) & (self._count >= 2)).float().reshape(-1)     # nothing to sample from < 2 episodes
        t = start.unsqueeze(-1) + self._win                                  # [B,W,n]
        e = ep.view(-1, 1, 1).expand(-1, W, n)
        g = lambda x: x[e, t].reshape(batch_size * W, n, *x.shape[2:])
        return g(self.ep_state), g(self.ep_action), g(self.ep_next), g(self.ep_reward), g(self.ep_not_done), weight
