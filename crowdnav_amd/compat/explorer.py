"""Explorer — the reference's episode driver surface (crowd_nav/utils/explorer.py:7-125).

run_k_episodes(k, phase, ...) keeps the reference's signature, bookkeeping and log lines.  When the robot's
policy lives on the device (ORCA) and no replay memory has to be filled, all k episodes run as ONE batch of
min(k, max_envs) envs inside the fused rollout kernel (cn_rollout): episode i of the call is the scenario the
reference would have produced on its i-th env.reset(phase).  Otherwise the reference's own loop runs on top of
CrowdSim.step (one launch per transition)."""
import copy
import logging
import os

import torch

from .. import _lib
from ..engine import BatchedCrowdSim
from .policy import is_device_orca
from .sarl import SARL
from ..sarl_rollout import SarlRollout
from .types import Collision, Danger, ReachGoal, Timeout


def average(values):
    return sum(values) / len(values) if values else 0


class Explorer(object):
    max_envs = 4096  # envs per batched launch

    def __init__(self, env, robot, device, memory=None, gamma=None, target_policy=None):
        self.env = env
        self.robot = robot
        self.device = device
        self.memory = memory
        self.gamma = gamma
        self.target_policy = target_policy
        self.target_model = None
        self.last_batch = None  # per-episode arrays of the last batched call (for callers that want more)

    def update_target_model(self, target_model):
        """explorer.py:26-27 (copy.deepcopy).  When a target network of the same architecture exists already, its parameters
        are overwritten in place: the same values, and the captured graph of its forward (_td_values) stays valid."""
        old = self.target_model
        if old is not None and type(old) is type(target_model):
            try:
                src, dst = target_model.state_dict(), old.state_dict()
                if src.keys() == dst.keys() and all(src[k].shape == dst[k].shape and src[k].dtype == dst[k].dtype and
                                                    src[k].device == dst[k].device for k in src):
                    old.load_state_dict(src)
                    old.train(target_model.training)  # (what copy.deepcopy would have carried over besides the parameters)
                    return
            except RuntimeError:
                pass
        self.target_model = copy.deepcopy(target_model)
        self._td_graph = None

    def _td_values(self, nxt):
        """target_model(next states) for the TD targets of update_memory (explorer.py:113-116), flat.  On a GPU the forward — some
        35 tiny kernels for a few dozen rows: launch-bound — is replayed from a hipGraph captured on a fixed number of rows (the
        rows beyond the call's hold earlier, finite inputs and are not read back).  CROWDNAV_AMD_TD_GRAPH=0: always eager."""
        model = self.target_model
        # (module, name) of every parameter, walked once per model object: model.parameters() visits every submodule — twice per
        # sampled episode it was ~0.09 ms of host time; the Parameter objects are looked up afresh below, so a replaced or moved one
        # still changes the key
        slots = getattr(self, '_td_slots', (None, None))
        if slots[0] is None or slots[0]() is not model:
            import weakref
            slots = self._td_slots = (weakref.ref(model), [(m, k) for m in model.modules() for k, p in m._parameters.items()
                                                            if p is not None])
        live = [m._parameters[k] for m, k in slots[1]]
        x = nxt.to(live[0].device)
        v = self._td_values_on_engine(model, live, x)
        if v is not None:
            return v
        n = int(x.shape[0])
        if (not x.is_cuda or n == 0 or getattr(self, '_td_graph_failed', False)
                or os.environ.get('CROWDNAV_AMD_TD_GRAPH', '1') == '0'):
            return model(x).reshape(-1)
        g = getattr(self, '_td_graph', None)
        # the graph replays reads of the parameters' STORAGE: a model whose parameters moved (.to(), .half(), load_state_dict(
        # assign=True), another module at a recycled id) must be captured again, not replayed on the old weights
        key = (id(model), tuple(x.shape[1:]), x.dtype, tuple(p.data_ptr() for p in live), model.training)
        if g is None or g['key'] != key or g['x'].shape[0] < n:
            rows = max(128, 2 * n if g is not None and g['key'] == key else n)
            try:
                sx = torch.zeros((rows,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side), torch.no_grad():
                    for _ in range(2):
                        model(sx)
                torch.cuda.current_stream().wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.no_grad(), torch.cuda.graph(graph):
                    sy = model(sx)
                g = self._td_graph = dict(key=key, x=sx, y=sy, graph=graph)
            except Exception as exc:  # noqa: BLE001 - e.g. a layer whose library call cannot be captured: run eagerly from now on
                logging.warning('TD-target forward: graph capture failed (%s); running eagerly', exc)
                self._td_graph, self._td_graph_failed = None, True
                return model(x).reshape(-1)
        g['x'][:n].copy_(x)
        g['graph'].replay()
        return g['y'][:n].reshape(-1)  # (a view of the graph's output: consumed before the next replay — update_memory converts it at once)

    def _td_values_on_engine(self, model, live, x):
        """The target network's forward by the library's own network kernel (cn_sarl_values: ONE launch on the narrow tiles, the
        rows read where they lie) on an engine that only holds the target's weights — uploaded again whenever a parameter's
        version counter or address moved (update_target_model's in-place copy bumps the versions).  None: not this configuration
        (another policy, occupancy maps, CPU, CROWDNAV_AMD_TD_KERNEL=0) — the caller runs the framework's forward."""
        policy = self.robot.policy if self.robot is not None else None
        cfg = getattr(policy, 'net_cfg', None)
        if (cfg is None or not x.is_cuda or x.dim() != 3 or x.shape[2] != 13 or x.dtype != torch.float32 or x.shape[0] == 0
                or cfg.get('model', 'sarl') not in ('sarl', 'lstm_rl') or cfg.get('with_om') or cfg.get('interaction_dims')
                or getattr(self, '_td_kernel_off', False) or os.environ.get('CROWDNAV_AMD_TD_KERNEL', '1') == '0'
                or type(model) is not type(getattr(policy, 'model', None)) or getattr(self, '_rl_engine_cache', None) is None):
            return None
        n, H = int(x.shape[0]), int(x.shape[1])
        K = len(policy.action_space)
        envs = 2
        while envs * K < n:
            envs *= 2
        if envs > 8:  # (more rows than the narrow tiles take — one workgroup per CU: this call's forward is the framework's)
            return None
        sig = tuple((p.data_ptr(), p._version) for p in live)
        cached = getattr(self, '_td_engine', None)
        try:
            if cached is None or cached['model']() is not model or cached['envs'] < envs or cached['H'] != H:
                import weakref
                base = dict(self._rl_engine_cache[0][0])  # the sampling engine's configuration: the same crowd, robot, widths
                if base['num_humans'] != H:
                    return None
                base['num_envs'] = envs
                eng = BatchedCrowdSim(**base)
                eng.sarl_configure(**policy.engine_kwargs())
                cached = self._td_engine = dict(model=weakref.ref(model), envs=envs, H=H, eng=eng, sig=None)
            if cached['sig'] != sig:
                cached['eng'].sarl_set_weights({k: m._parameters[k_] for (m, k_), k in
                                                zip(self._td_slots[1], self._td_names(model))})
                cached['sig'] = sig
            return cached['eng'].sarl_values(x.contiguous())
        except _lib.CrowdNavAmdError as exc:  # e.g. CN_ERR_UNSUPPORTED for widths / sizes off the narrow tiles
            logging.info('TD targets: cn_sarl_values not available here (%s); using the framework forward', exc)
            self._td_kernel_off = True
            return None

    def _td_names(self, model):
        """state_dict names of the parameter slots of _td_values, in the same order (module walk: prefix + parameter name)."""
        names = getattr(self, '_td_name_cache', (None, None))
        if names[0] is None or names[0]() is not model:
            import weakref
            names = self._td_name_cache = (weakref.ref(model), [(prefix + '.' if prefix else '') + k
                                                               for prefix, m in model.named_modules()
                                                               for k, p in m._parameters.items() if p is not None])
        return names[1]

    # ------------------------------------------------------------------ explorer.py:21-90
    def run_k_episodes(self, k, phase, update_memory=False, imitation_learning=False, episode=None,
                       print_failure=False):
        self.robot.policy.set_phase(phase)
        on_device = is_device_orca(self.robot.policy) or (isinstance(self.robot.policy, SARL) and phase != 'train')
        batched = (on_device and not update_memory and hasattr(self.env, 'engine_config')
                   and self.env.case_counter[phase] >= 0
                   and self.env.case_counter[phase] + k <= self.env.case_size[phase])  # no wrap of the case table
        no_wrap = (self.env.case_counter[phase] >= 0 and self.env.case_counter[phase] + k <= self.env.case_size[phase])
        batched_il = (is_device_orca(self.robot.policy) and update_memory and imitation_learning and no_wrap
                      and hasattr(self.env, 'engine_config') and isinstance(self.target_policy, SARL)
                      and getattr(self.target_policy, 'kinematics', 'holonomic') == 'holonomic')
        batched_rl = (isinstance(self.robot.policy, SARL) and phase == 'train' and update_memory
                      and not imitation_learning and no_wrap and hasattr(self.env, 'engine_config')
                      and getattr(self.robot.policy, 'env', None) is self.env)
        value_net = isinstance(self.robot.policy, SARL) or isinstance(self.target_policy, SARL)
        if value_net and update_memory and self._scenario_of(phase)[1] == 'mixed' and hasattr(self.env, 'engine_config'):
            # Acting under the mixed rule works (the kernels mask an episode's absent humans); FILLING A REPLAY MEMORY does
            # not: the states would hold a different number of humans per episode, which the reference cannot batch either
            # (its DataLoader stacks them; train.config keeps train_val_sim = circle_crossing).
            raise NotImplementedError('replay states under the mixed rule are ragged (a different number of humans per '
                                      'episode): train on circle_crossing / square_crossing as the reference does')
        if batched_il:
            stats = self._run_batched_imitation(k, phase)
        elif batched_rl:
            stats = self._run_batched_rl(k, phase)
        elif batched:
            stats = self._run_batched(k, phase)
        else:
            stats = self._run_sequential(k, phase, update_memory, imitation_learning)
        self._report(k, phase, episode, print_failure, *stats)

    def _run_sequential(self, k, phase, update_memory, imitation_learning):
        success_times, collision_times, timeout_times = [], [], []
        too_close, min_dist, cumulative_rewards = 0, [], []
        collision_cases, timeout_cases = [], []
        for i in range(k):
            ob = self.env.reset(phase)
            done = False
            states, actions, rewards = [], [], []
            while not done:
                action = self.robot.act(ob)
                ob, reward, done, info = self.env.step(action)
                states.append(self.robot.policy.last_state)
                actions.append(action)
                rewards.append(reward)
                if isinstance(info, Danger):
                    too_close += 1
                    min_dist.append(info.min_dist)
            if isinstance(info, ReachGoal):
                success_times.append(self.env.global_time)
            elif isinstance(info, Collision):
                collision_cases.append(i)
                collision_times.append(self.env.global_time)
            elif isinstance(info, Timeout):
                timeout_cases.append(i)
                timeout_times.append(self.env.time_limit)
            else:
                raise ValueError('Invalid end signal from environment')
            if update_memory and isinstance(info, (ReachGoal, Collision)):
                self.update_memory(states, actions, rewards, imitation_learning)
            cumulative_rewards.append(sum([pow(self.gamma, t * self.robot.time_step * self.robot.v_pref) * reward
                                           for t, reward in enumerate(rewards)]))
        return (success_times, collision_times, timeout_times, collision_cases, timeout_cases, too_close,
                average(min_dist), cumulative_rewards)

    def _run_batched(self, k, phase):
        env = self.env
        self.robot.time_step = env.time_step  # CrowdSim.reset does this (crowd_sim.py:296-298)
        self.robot.policy.time_step = env.time_step
        human_num, rule, offset = self._scenario_of(phase)
        start = env.case_counter[phase]
        size = env.case_size[phase]
        B = int(min(k, self.max_envs))
        per_env = (k + B - 1) // B
        max_steps = int(round(env.time_limit / env.time_step)) + 2
        names = ('ep_outcome', 'ep_steps', 'ep_return', 'ep_time', 'ep_danger', 'ep_danger_dmin_sum')
        # episode i of this call is case start + i of the phase (no wrap: checked by the caller)
        if is_device_orca(self.robot.policy):
            eng = BatchedCrowdSim(**env.engine_config(B, human_num, rule, _lib.ROBOT_ORCA))
            eng.set_gamma(self.gamma)
            self._share_robot_sim(eng, human_num, rule, offset + start)
            # (no job-wide counter, no in-kernel statistics: the records are read once below, as explorer.py:50-90 does)
            bufs = eng.rollout_begin(seed_base=offset + start, seed_mod=size, episode_limit=k, record_capacity=per_env,
                                     per_env_transitions=True)
            while True:
                eng.rollout(max_steps)
                if int(bufs['active'].sum().item()) == 0:
                    break
            rec = {n: bufs[n].cpu().numpy() for n in names}
        else:  # SARL value network: select + step + masked reset per batched step
            policy = self.robot.policy
            if policy.action_space is None:
                policy.build_action_space(self.robot.v_pref)
            eng = BatchedCrowdSim(**env.engine_config(B, human_num, rule, _lib.ROBOT_EXTERNAL))
            eng.sarl_configure(**policy.engine_kwargs())
            eng.sarl_set_weights(policy.model.state_dict())
            ro = SarlRollout(eng, self.gamma, seed_base=offset + start, seed_mod=size, episode_limit=k,
                             record_capacity=per_env)
            while ro.any_active():
                ro.run(8)
            rec = {n: ro.rec[m].cpu().numpy() for n, m in zip(names, ('outcome', 'steps', 'ret', 'time', 'danger', 'dsum'))}
        env.case_counter[phase] = (start + k) % size
        # episode id c = b + j*B  ->  record [b, j]
        order = [(c % B, c // B) for c in range(k)]
        outcome = [int(rec['ep_outcome'][b, j]) for b, j in order]
        times = [float(rec['ep_time'][b, j]) for b, j in order]
        returns = [float(rec['ep_return'][b, j]) for b, j in order]
        self.last_batch = dict(outcome=outcome, nav_time=times, discounted_return=returns,
                               steps=[int(rec['ep_steps'][b, j]) for b, j in order])
        success_times = [t for o, t in zip(outcome, times) if o == _lib.REACH_GOAL]
        collision_times = [t for o, t in zip(outcome, times) if o == _lib.COLLISION]
        timeout_times = [t for o, t in zip(outcome, times) if o == _lib.TIMEOUT]
        collision_cases = [i for i, o in enumerate(outcome) if o == _lib.COLLISION]
        timeout_cases = [i for i, o in enumerate(outcome) if o == _lib.TIMEOUT]
        too_close = int(sum(rec['ep_danger'][b, j] for b, j in order))
        dsum = float(sum(rec['ep_danger_dmin_sum'][b, j] for b, j in order))
        return (success_times, collision_times, timeout_times, collision_cases, timeout_cases, too_close,
                dsum / too_close if too_close else 0, returns)

    def _run_batched_imitation(self, k, phase):
        """Imitation-learning data collection (train.py:115-129): k ORCA-robot episodes in lock step on the device,
        then update_memory(..., imitation_learning=True) for all of them at once.  Per step cn_sarl_transform writes
        the target policy's transform of the joint state the ORCA robot saw (explorer.py:99:
        target_policy.transform(state), humans in env order) straight into a [B, T, H, D] trajectory tensor; values
        are the discounted Monte-Carlo returns of explorer.py:100-105 (host float64, the reference's left-to-right
        sum); (state, value) pairs enter the memory in the reference's order."""
        import numpy as np
        env, policy = self.env, self.target_policy
        self.robot.time_step = env.time_step
        self.robot.policy.time_step = env.time_step
        if policy.action_space is None:
            policy.build_action_space(self.robot.v_pref)
        human_num, rule, offset = self._scenario_of(phase)
        start, dt, vp = env.case_counter[phase], env.time_step, self.robot.v_pref
        max_steps = int(round(env.time_limit / dt)) + 2
        D = policy.input_dim()
        single = policy.net_cfg.get('model') == 'cadrl'  # CADRL.transform (cadrl.py:174-185): one human, [13]
        outcome, length, rewards_all, danger_n, danger_sum = [], [], [], 0, 0.0
        for c0 in range(0, k, self.max_envs):
            B = min(self.max_envs, k - c0)
            eng = BatchedCrowdSim(**env.engine_config(B, human_num, rule, _lib.ROBOT_ORCA))
            eng.sarl_configure(**policy.engine_kwargs())  # only the transform runs on this engine
            self._share_robot_sim(eng, human_num, rule, offset + start)
            eng.reset(offset + start + c0 + np.arange(B))
            traj = torch.zeros(B, max_steps, human_num, D, dtype=torch.float32, device=eng.device)
            hist_r, hist_i, hist_d = [], [], []
            alive = torch.ones(B, dtype=torch.bool, device=eng.device)
            for t in range(max_steps):
                eng.sarl_transform(out=traj[:, t], env_stride=max_steps * human_num * D, sort_humans=False)
                out = eng.step(None, update=True, want_obs=False)
                hist_r.append(out['reward'])
                hist_i.append(out['info'])
                hist_d.append(out['dmin'])
                alive = alive & (out['done'] == 0)
                if t % 8 == 7 and not bool(alive.any().item()):
                    break
            R = torch.stack(hist_r).cpu().numpy()      # [T, B]
            I = torch.stack(hist_i).cpu().numpy()
            Dm = torch.stack(hist_d).cpu().numpy()
            terminal = I >= _lib.REACH_GOAL
            if not terminal.any(axis=0).all():
                raise ValueError('Invalid end signal from environment')
            Tb = terminal.argmax(axis=0) + 1           # first terminal step of every episode
            last = I[Tb - 1, np.arange(B)]
            for b in range(B):
                n = int(Tb[b])
                outcome.append(int(last[b]))
                length.append(n)
                rewards_all.append(R[:n, b].tolist())
                dang = I[:n, b] == _lib.DANGER
                danger_n += int(dang.sum())
                danger_sum += float(Dm[:n, b][dang].sum())
            # explorer.py:66-69, 92-125 for every ReachGoal / Collision episode of this batch
            keep = np.flatnonzero((last == _lib.REACH_GOAL) | (last == _lib.COLLISION))
            if len(keep):
                if self.memory is None or self.gamma is None:
                    raise ValueError('Memory or gamma value is not set!')
                b_idx = np.repeat(keep, Tb[keep])
                i_idx = np.concatenate([np.arange(Tb[b]) for b in keep])
                x = traj[torch.as_tensor(b_idx, device=eng.device), torch.as_tensor(i_idx, device=eng.device)]
                values = []
                for b in keep:
                    rw = rewards_all[c0 + int(b)]
                    for i in range(len(rw)):
                        values.append(sum([pow(self.gamma, max(t - i, 0) * dt * vp) * r * (1 if t >= i else 0)
                                           for t, r in enumerate(rw)]))
                self._push_all(x[:, 0] if single else x, torch.Tensor(values))
        env.case_counter[phase] = (start + k) % env.case_size[phase]

        self.last_batch = dict(outcome=outcome, steps=length, env_steps=int(sum(length)))
        times = [length[e] * dt for e in range(k)]
        success_times = [times[e] for e in range(k) if outcome[e] == _lib.REACH_GOAL]
        collision_times = [times[e] for e in range(k) if outcome[e] == _lib.COLLISION]
        timeout_times = [env.time_limit for e in range(k) if outcome[e] == _lib.TIMEOUT]
        collision_cases = [e for e in range(k) if outcome[e] == _lib.COLLISION]
        timeout_cases = [e for e in range(k) if outcome[e] == _lib.TIMEOUT]
        returns = [sum([pow(self.gamma, t * dt * vp) * r for t, r in enumerate(rw)]) for rw in rewards_all]
        return (success_times, collision_times, timeout_times, collision_cases, timeout_cases, danger_n,
                danger_sum / danger_n if danger_n else 0, returns)

    def _share_robot_sim(self, eng, human_num, rule, first_seed):
        """One persistent ORCA policy object = one captured rvo2 simulator (orca.py:95-110): every env of a batched run
        sees the radii the policy captured at its first episode — this call's first episode if it has none yet."""
        env = self.env
        if not (env.randomize_attributes and is_device_orca(self.robot.policy)):
            return  # constant radii: every capture is the same
        cap = getattr(self.robot.policy, '_rsim', None)
        if cap is None or len(cap[0]) != human_num + 1:
            one = BatchedCrowdSim(**env.engine_config(1, human_num, rule, _lib.ROBOT_ORCA))
            one.reset([first_seed])
            cap = env.robot_sim_capture(one.get_state()[0].cpu().numpy()[0, :, 6].tolist())
        eng.set_robot_sim(*cap)

    def _scenario_of(self, phase):
        env = self.env
        multi = getattr(self.robot.policy, 'multiagent_training', None)
        if phase == 'test':
            human_num, rule = env.human_num, env.test_sim
        else:  # crowd_sim.py:266-267, 277-279
            human_num, rule = (env.human_num if multi else 1), ('circle_crossing' if not multi else env.train_val_sim)
        offset = {'train': env.case_capacity['val'] + env.case_capacity['test'], 'val': 0,
                  'test': env.case_capacity['val']}[phase]
        return human_num, rule, offset

    def _rl_engine(self, B, human_num, rule):
        """The batched engine of the RL sampling phase, kept between calls (train.py calls run_k_episodes once per
        training episode: 10 000 times in the shipped schedule)."""
        policy = self.robot.policy
        # fast path: everything engine_config reads, by value or (the config object, the policy, its action-space list) by
        # identity — the dict below with its two configparser reads and its sort was 0.03 ms of a sampled episode, 0.1 ms behind
        # the schedule's SGD batches when the interpreter's own data is cold
        env, robot = self.env, self.robot
        quick = (B, human_num, rule, env.time_step, env.time_limit, env.success_reward, env.collision_penalty, env.discomfort_dist,
                 env.discomfort_penalty_factor, robot.visible, getattr(policy, 'safety_space', 0), env.circle_radius,
                 env.square_width, robot.radius, robot.v_pref, env.randomize_attributes, env.device,
                 getattr(robot, 'kinematics', 'holonomic'), id(env.config), id(policy), id(policy.action_space))
        hit = getattr(self, '_rl_engine_quick', None)
        if hit is not None and hit[0] == quick and getattr(self, '_rl_engine_cache', None) is not None:
            return self._rl_engine_cache[1]
        cfg = self.env.engine_config(B, human_num, rule, _lib.ROBOT_EXTERNAL)
        # (the action table by the identity of the policy's action_space list — rebuilt tables are new lists; the cache entry holds
        # the list, so its id cannot be recycled — instead of 81 tuples converted and hashed per sampled episode)
        key = (tuple(sorted(cfg.items())), id(policy), id(policy.action_space))
        cached = getattr(self, '_rl_engine_cache', None)
        if cached is None or cached[0] != key:
            eng = BatchedCrowdSim(**cfg)
            eng.sarl_configure(**policy.engine_kwargs())
            self._rl_engine_cache = cached = (key, eng, policy.action_space)
        self._rl_engine_quick = (quick, env.config)  # (holds the config object: its id cannot be recycled)
        return cached[1]

    def _push_all(self, states, values):
        """(state, value) pairs into the replay memory in the given order (explorer.py:125)."""
        if hasattr(self.memory, 'push_batch'):
            self.memory.push_batch(states, values)
        else:
            states, values = states.to(self.device), values.to(self.device)
            for j in range(states.shape[0]):
                self.memory.push((states[j], values[j].reshape(1)))

    def _run_batched_rl(self, k, phase):
        """RL-phase sampling (train.py:147-157: run_k_episodes(sample_episodes, 'train', update_memory=True)) with the
        epsilon-greedy value-network robot (SARL, CADRL or LSTM-RL), k episodes in lock step on the device.  Per
        batched step:
        cn_sarl_select (greedy action of every env) -> cn_sarl_explore (the epsilon branch of
        multi_human_rl.py:28-31 on each env's own numpy stream, continued after its scenario draws) ->
        cn_sarl_transform (policy.last_state, written straight into the trajectory tensor) -> cn_step.  Then
        update_memory (explorer.py:92-125) for all ReachGoal / Collision episodes at once: the TD targets
        r + gamma^(dt v_pref) * target_model(next state) come from ONE batched forward of the target network, and the
        (state, value) pairs enter the memory in the reference's order (episode by episode, step by step)."""
        import numpy as np
        env, policy = self.env, self.robot.policy
        if self.memory is None or self.gamma is None:
            raise ValueError('Memory or gamma value is not set!')
        if policy.epsilon is None:
            raise AttributeError('Epsilon attribute has to be set in training phase')
        self.robot.time_step = env.time_step  # CrowdSim.reset does this (crowd_sim.py:296-298)
        policy.time_step = env.time_step
        if policy.action_space is None:
            policy.build_action_space(self.robot.v_pref)
        human_num, rule, offset = self._scenario_of(phase)
        start, dt, vp = env.case_counter[phase], env.time_step, self.robot.v_pref
        max_steps = int(round(env.time_limit / dt)) + 2
        gamma_bar = pow(self.gamma, dt * vp)
        D = policy.input_dim()
        outcome, length, returns, danger_n, danger_sum = [], [], [], 0, 0.0
        n_steps, actions_taken = 0, []
        for c0 in range(0, k, self.max_envs):
            B = min(self.max_envs, k - c0)
            prof = getattr(self, 'rl_profile', None)  # dict: seconds per part of a sampling call (synchronising: debugging only)
            if prof is not None:
                import time
                torch.cuda.synchronize()
                tp = [time.perf_counter()]

                def lap(name):
                    torch.cuda.synchronize()
                    tp.append(time.perf_counter())
                    prof[name] = prof.get(name, 0.0) + tp[-1] - tp[-2]
            else:
                lap = lambda name: None  # noqa: E731
            eng = self._rl_engine(B, human_num, rule)
            # (the parameters themselves, name -> Parameter, cached per model object: state_dict() builds 22 detached views per
            # sampled episode, ~0.09 ms of host time; the optimizer updates the same tensors in place, and a model that moves or
            # reloads keeps its Parameter objects — their addresses are read afresh by every call)
            wkey = getattr(self, '_rl_params', (None, None))
            if wkey[0] is None or wkey[0]() is not policy.model:
                import weakref
                wkey = self._rl_params = (weakref.ref(policy.model), dict(policy.model.named_parameters()))
            lap('  (engine lookup)')
            eng.sarl_set_weights(wkey[1])
            lap('  (weight re-pack)')
            # the seeds go up from a pinned buffer behind the weight re-pack, without a synchronisation (engine.reset waits
            # for the scenarios: ~0.1 ms per sampled episode of device idle time in front of the first step)
            skey = (id(eng), B)
            if getattr(self, '_rl_seeds', (None,))[0] != skey:
                self._rl_seeds = (skey, torch.empty(B, dtype=torch.int32).pin_memory(), torch.empty(B, dtype=torch.int32, device=eng.device))
            _, seeds_host, seeds_dev = self._rl_seeds
            seeds_host.numpy()[:] = (offset + start + c0 + np.arange(B)).astype(np.uint32).view(np.int32)
            with torch.cuda.stream(eng._stream):
                seeds_dev.copy_(seeds_host, non_blocking=True)
            eng.reset_async(seeds_dev, None)
            lap('  (seeds + reset)')
            # the histories live as long as the engine (one allocation + fill per shape, not five per sampled episode); every row
            # that is read below has been written by this call's steps, except traj's row T, which only feeds a value that
            # torch.where discards (stale rows are finite)
            # A few envs (the two-launch route of cn_sarl_sample_step: train.py samples ONE episode per call): ALL four histories
            # live in pinned host memory — the kernels only write them, a step's reward / min distance / action before its info
            # code — so that the host reads an episode's rows the moment its end code has arrived: no copy back, no wait for the
            # steps issued past the end, and the TD targets, the push and the host's statistics overlap.
            pin = B <= 8 and os.environ.get('CROWDNAV_AMD_RL_PINNED', '1') != '0'
            hkey = (id(eng), B, max_steps, human_num, D, pin)
            if getattr(self, '_rl_hist', (None,))[0] != hkey:
                z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=eng.device)  # noqa: E731
                # reward / min-distance / action / info histories are four regions of ONE byte buffer: one blocking copy
                # brings all of them to the host (four of them were a tenth of a millisecond per sampled episode)
                n = max_steps * B
                packed = torch.zeros((20 * n,), dtype=torch.uint8).pin_memory() if pin else z((20 * n,), torch.uint8)
                rew_, dmn_ = packed[:8 * n].view(torch.float64).view(max_steps, B), packed[8 * n:16 * n].view(torch.float64).view(max_steps, B)
                act_ = packed[16 * n:20 * n].view(torch.int32).view(max_steps, B)
                # the info codes go to PINNED host memory: the kernels only write them, and the host watches the episode ends
                # arrive while it keeps issuing steps — no device synchronisation inside an episode (round 6; a check every 8
                # steps was a bubble of ~40 us each time and 3.5 wasted steps per episode on average)
                inf_ = torch.empty((max_steps, B), dtype=torch.uint8).pin_memory()
                self._rl_hist = (hkey, z((B, max_steps, human_num, D), torch.float32), rew_, inf_, dmn_, act_,
                                 z((B,), torch.uint8), z((B,), torch.uint8), z((B, 2), torch.float64), packed)
            _, traj, rew, inf, dmn, act, alive, done, action, packed = self._rl_hist
            alive.fill_(1)
            done.zero_()
            kUnwritten = 255
            inf_np = inf.numpy()
            inf_np.fill(kUnwritten)
            T = 0
            lap('weights + reset')
            # Per step: ONE library call (cn_sarl_sample_step: two launches at one env) and no torch kernel — every result
            # lands in its row of the histories, and an env leaves `alive` at the start of the step after its episode ended.
            # The host runs at most `ahead` steps in front of the device (the rows of info it has seen arrive tell it where the
            # device is) and stops issuing once every env's episode-end code is there; the steps already issued for an env
            # that has finished are skipped by the kernels (two-launch route) or step a retired env (general route).
            step = eng.sarl_sampler(traj, rew, inf, dmn, act, alive, done, action)
            fused_before = eng.launch_counts()['sarl_decide_steps']
            eps = float(policy.epsilon)
            ahead, seen, spins = int(os.environ.get('CROWDNAV_AMD_RL_AHEAD', '2')), 0, 0
            finished = np.zeros(B, dtype=bool)
            for t in range(max_steps):
                step(t, eps)
                T = t + 1
                while seen < T:  # rows the device has completed: every env that still samples has written its code
                    row = inf_np[seen]
                    if ((row != kUnwritten) | finished).all():
                        finished |= (row != kUnwritten) & (row >= _lib.REACH_GOAL)
                        seen += 1
                        spins = 0
                    elif T - seen > ahead and spins < 20000000:
                        spins += 1   # (bounded: a device error surfaces at eng.sync() below instead of hanging here)
                    else:
                        break
                if finished.all():
                    break
            # pinned histories are complete up to every env's end code once that code is there — provided the steps ran the
            # two-launch route (its last kernel writes a step's outputs in that order; launch counters: host-side, no device
            # work); anything else waits for the device as before
            fused_steps = eng.launch_counts()['sarl_decide_steps'] - fused_before
            if not (pin and fused_steps == T and finished.all()):
                eng.sync()
            lap('steps')
            if prof is not None:
                prof['n_steps_issued'] = prof.get('n_steps_issued', 0) + T
            host, n = (packed if pin else packed.cpu()).numpy(), max_steps * B
            lap('  (histories to the host)')
            R, Dm = host[:8 * n].view(np.float64).reshape(max_steps, B)[:T], host[8 * n:16 * n].view(np.float64).reshape(max_steps, B)[:T]
            Ac, I = host[16 * n:20 * n].view(np.int32).reshape(max_steps, B)[:T], inf_np[:T].copy()
            I[I == kUnwritten] = _lib.NOTHING   # (rows behind an env's last step on the two-launch route)
            terminal = I >= _lib.REACH_GOAL
            if not terminal.any(axis=0).all():
                raise ValueError('Invalid end signal from environment')
            Tb = terminal.argmax(axis=0) + 1                                   # steps of every episode
            if ((Ac == -2) & (np.arange(T)[:, None] < Tb[None, :])).any():    # greedy branch without a finite value
                raise ValueError('Value network is not well trained. ')       # multi_human_rl.py:57-58
            n_steps += int(Tb.sum())
            last = I[Tb - 1, np.arange(B)]
            keep = np.flatnonzero((last == _lib.REACH_GOAL) | (last == _lib.COLLISION))
            if len(keep):
                # rows in push order: episode by episode, step by step.  An episode's rows are slices of the histories (no
                # index tensors to upload, no gathers): states [0, n), next states [1, n] — row n only feeds the value that the
                # last step replaces by its reward
                single = policy.net_cfg.get('model') == 'cadrl'      # CADRL.transform: one human, [13]
                ns = [int(Tb[b]) for b in keep]
                if max(ns) < max_steps:
                    cat = lambda parts: parts[0] if len(parts) == 1 else torch.cat(parts)  # noqa: E731
                    states = cat([traj[b, :n_] for b, n_ in zip(keep, ns)])               # [N, H, D]
                    nxt = cat([traj[b, 1:n_ + 1] for b, n_ in zip(keep, ns)])
                    r = cat([rew[:n_, b] for b, n_ in zip(keep, ns)])
                    ends = np.cumsum(ns) - 1                                              # the last step of every episode
                    if pin:  # (a slice of the pinned history: goes up behind the steps, asynchronously)
                        r = r.to(eng.device, non_blocking=True)
                else:  # (an episode as long as the histories: its row n does not exist)
                    b_idx = np.repeat(keep, Tb[keep])
                    i_idx = np.concatenate([np.arange(Tb[b]) for b in keep])
                    bt = torch.as_tensor(b_idx, device=eng.device)
                    it = torch.as_tensor(i_idx, device=eng.device)
                    states, nxt = traj[bt, it], traj[bt, torch.clamp(it + 1, max=max_steps - 1)]
                    r = torch.from_numpy(R[i_idx, b_idx]).to(eng.device) if pin else rew[it, bt]
                    ends = np.flatnonzero(i_idx == Tb[b_idx] - 1)
                if single:
                    states, nxt = states[:, 0], nxt[:, 0]
                lap('  (rows of the episode)')
                with torch.no_grad():
                    v_next = self._td_values(nxt).to(eng.device)
                    lap('  (target network)')
                    values = torch.add(r, v_next.double(), alpha=gamma_bar)   # float64, as the reference's Python floats
                    if len(ends) == 1:
                        values[-1:].copy_(r[-1:])
                    else:
                        e_idx = torch.as_tensor(ends, device=eng.device)
                        values[e_idx] = r[e_idx]
                self._push_all(states, values.float())
            lap('read-back + TD targets + push')
            for b in range(B):
                n = int(Tb[b])
                outcome.append(int(last[b]))
                length.append(n)
                rw = R[:n, b].tolist()
                actions_taken.append(Ac[:n, b].tolist())
                returns.append(sum([pow(self.gamma, t * dt * vp) * r_ for t, r_ in enumerate(rw)]))
                dang = I[:n, b] == _lib.DANGER
                danger_n += int(dang.sum())
                danger_sum += float(Dm[:n, b][dang].sum())
        env.case_counter[phase] = (start + k) % env.case_size[phase]
        self.last_batch = dict(outcome=outcome, steps=length, discounted_return=returns,
                               nav_time=[n * dt for n in length], env_steps=n_steps, actions=actions_taken)
        times = [n * dt for n in length]
        success_times = [times[e] for e in range(k) if outcome[e] == _lib.REACH_GOAL]
        collision_times = [times[e] for e in range(k) if outcome[e] == _lib.COLLISION]
        timeout_times = [env.time_limit for e in range(k) if outcome[e] == _lib.TIMEOUT]
        collision_cases = [e for e in range(k) if outcome[e] == _lib.COLLISION]
        timeout_cases = [e for e in range(k) if outcome[e] == _lib.TIMEOUT]
        return (success_times, collision_times, timeout_times, collision_cases, timeout_cases, danger_n,
                danger_sum / danger_n if danger_n else 0, returns)

    def _report(self, k, phase, episode, print_failure, success_times, collision_times, timeout_times,
                collision_cases, timeout_cases, too_close, avg_min_dist, cumulative_rewards):
        success, collision, timeout = len(success_times), len(collision_times), len(timeout_times)
        success_rate = success / k
        collision_rate = collision / k
        assert success + collision + timeout == k
        avg_nav_time = sum(success_times) / len(success_times) if success_times else self.env.time_limit
        extra_info = '' if episode is None else 'in episode {} '.format(episode)
        logging.info('{:<5} {}has success rate: {:.2f}, collision rate: {:.2f}, nav time: {:.2f}, total reward: {:.4f}'.
                     format(phase.upper(), extra_info, success_rate, collision_rate, avg_nav_time,
                            average(cumulative_rewards)))
        if phase in ['val', 'test']:
            num_step = sum(success_times + collision_times + timeout_times) / self.robot.time_step
            logging.info('Frequency of being in danger: %.2f and average min separate distance in danger: %.2f',
                         too_close / num_step, avg_min_dist)
        if print_failure:
            logging.info('Collision cases: ' + ' '.join([str(x) for x in collision_cases]))
            logging.info('Timeout cases: ' + ' '.join([str(x) for x in timeout_cases]))
        self.last_stats = dict(success_rate=success_rate, collision_rate=collision_rate, nav_time=avg_nav_time,
                               total_reward=average(cumulative_rewards), too_close=too_close,
                               collision_cases=collision_cases, timeout_cases=timeout_cases)

    # ------------------------------------------------------------------ explorer.py:92-125
    def update_memory(self, states, actions, rewards, imitation_learning=False):
        """(state, value) pairs of one episode into the replay memory (explorer.py:92-125).  The TD targets of the RL
        branch need the target network's value of every next state: they are evaluated in ONE batched forward instead of
        one per step."""
        if self.memory is None or self.gamma is None:
            raise ValueError('Memory or gamma value is not set!')
        dt_vp = self.robot.time_step * self.robot.v_pref
        next_values = None
        if not imitation_learning and len(states) > 1:
            with torch.no_grad():
                next_values = self.target_model(torch.stack(list(states[1:]))).reshape(-1).tolist()
        for i, state in enumerate(states):
            reward = rewards[i]
            if imitation_learning:
                state = self.target_policy.transform(state)
                value = sum([pow(self.gamma, max(t - i, 0) * dt_vp) * reward * (1 if t >= i else 0)
                             for t, reward in enumerate(rewards)])
            elif i == len(states) - 1:
                value = reward  # terminal state
            else:
                value = reward + pow(self.gamma, dt_vp) * next_values[i]
            self.memory.push((state, torch.Tensor([value]).to(self.device)))
