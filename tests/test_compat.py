"""The drop-in surface (crowdnav_amd.compat): gym-style CrowdSim + Explorer against the reference-generated
fixtures.  CPU part: value types, config plumbing, report formatting.  GPU part: trajectories and the
Explorer.run_k_episodes log line of `test.py --policy orca` (0.43 / 0.57 / 10.86)."""
import logging

import numpy as np
import pytest

from conftest import episodes_of, load_golden


def _setup(robot_visible=False, overrides=None):
    import crowdnav_amd.compat as c
    ov = {('robot', 'visible'): 'true' if robot_visible else 'false'}
    ov.update(overrides or {})
    cfg = c.default_env_config(ov)
    env = c.CrowdSim()
    env.configure(cfg)
    robot = c.Robot(cfg, 'robot')
    policy = c.ORCA()
    robot.set_policy(policy)
    env.set_robot(robot)
    policy.set_env(env)
    return c, env, robot


def test_types_and_config_cpu():
    c, env, robot = _setup()
    assert env.case_size == {'train': 2 ** 32 - 1 - 2000, 'val': 100, 'test': 500}
    assert env.case_capacity['val'] + env.case_capacity['test'] == 2000
    assert (env.time_limit, env.time_step, env.human_num) == (25, 0.25, 5)
    o = c.ObservableState(1.0, 2.0, 3.0, 4.0, 0.3)
    f = c.FullState(1.0, 2.0, 3.0, 4.0, 0.3, 5.0, 6.0, 1.0, 0.0)
    # joint row = self (9) followed by human (5), as multi_human_rl.py:43 flattens it
    assert (f + o) == (1.0, 2.0, 3.0, 4.0, 0.3, 5.0, 6.0, 1.0, 0.0) + (1.0, 2.0, 3.0, 4.0, 0.3)
    assert str(c.ReachGoal()) == 'Reaching goal' and str(c.Nothing()) == '' and c.Danger(0.1).min_dist == 0.1
    cfg = env.engine_config(7, 5, 'square_crossing', 1)
    assert cfg['num_envs'] == 7 and cfg['scenario_rule'] == 1 and cfg['robot_visible'] == 0
    mixed = env.engine_config(3, 2, 'mixed', 1)  # the rule draws its own number of humans: 5 slots whatever was asked
    assert mixed['scenario_rule'] == 2 and mixed['num_humans'] == 5
    with pytest.raises(NotImplementedError):
        env.engine_config(1, 5, 'trajnet', 1)
    with pytest.raises(AttributeError):
        c.CrowdSim().reset('test')


def test_report_line_format_cpu(caplog):
    c, env, robot = _setup()
    robot.time_step = 0.25
    ex = c.Explorer(env, robot, 'cpu', gamma=0.9)
    with caplog.at_level(logging.INFO):
        ex._report(4, 'test', None, True, [10.0, 11.0], [3.0], [25], [2], [3], 5, 0.081, [1.0, 0.5, -0.25, 0.0])
    assert ('TEST  has success rate: 0.50, collision rate: 0.25, nav time: 10.50, total reward: 0.3125'
            in caplog.text)
    assert 'Frequency of being in danger: 0.03 and average min separate distance in danger: 0.08' in caplog.text
    assert 'Collision cases: 2' in caplog.text and 'Timeout cases: 3' in caplog.text


@pytest.mark.gpu
@pytest.mark.parametrize('name,visible', [('traj_invisible_h5.npz', False), ('traj_visible_h5.npz', True)])
def test_gym_surface_reproduces_reference_episodes(name, visible):
    """reset / robot.act / step exactly as Explorer drives them, vs the unmodified reference."""
    c, env, robot = _setup(robot_visible=visible)
    g = load_golden(name)
    for case, e in zip(g['cases'].tolist()[:6], episodes_of(g)[:6]):
        ob = env.reset('test', case)
        assert len(ob) == 5 and abs(ob[0].px - e['states'][0][1][0]) <= 1e-12
        # start from the reference's exact initial state (numpy vs device cos/sin may differ in the last ulp)
        env._eng.set_state(e['states'][:1], np.zeros(1))
        env._pull()
        ob = [h.get_observable_state() for h in env.humans]
        for t in range(len(e['actions'])):
            action = robot.act(ob)
            assert (action.vx, action.vy) == tuple(e['actions'][t])
            look = env.onestep_lookahead(action)
            ob, reward, done, info = env.step(action)
            assert look[1] == reward and look[2] == done
            assert reward == e['rewards'][t] and done == bool(e['dones'][t])
            assert type(info).__name__ == ('Nothing', 'Danger', 'ReachGoal', 'Collision', 'Timeout')[e['infos'][t]]
            got = np.array([[h.px, h.py, h.vx, h.vy] for h in env.humans])
            assert np.array_equal(got, e['states'][t + 1][1:, :4])
            assert (robot.px, robot.py) == tuple(e['states'][t + 1][0, :2])
        assert done and env.global_time == 0.25 * len(e['actions'])
    assert env.case_counter['test'] == (g['cases'].tolist()[5] + 1) % 500


@pytest.mark.gpu
def test_explorer_500_test_cases_log_line(caplog):
    """`python test.py --policy orca` on the shipped configs: success 0.43, collision 0.57, nav time 10.86."""
    c, env, robot = _setup(robot_visible=False)
    ex = c.Explorer(env, robot, 'cuda:0', gamma=0.9)
    with caplog.at_level(logging.INFO):
        ex.run_k_episodes(env.case_size['test'], 'test', print_failure=True)
    assert 'TEST  has success rate: 0.43, collision rate: 0.57, nav time: 10.86, total reward: -0.0220' in caplog.text
    assert 'Frequency of being in danger: 0.30 and average min separate distance in danger: 0.08' in caplog.text
    g = load_golden('outcomes_500.npz')
    assert ex.last_batch['outcome'] == g['invisible_info'].tolist()
    assert ex.last_batch['steps'] == g['invisible_steps'].tolist()
    assert ex.last_stats['collision_cases'] == np.nonzero(g['invisible_info'] == 3)[0].tolist()
    assert env.case_counter['test'] == 0  # wrapped: 500 % 500


@pytest.mark.gpu
def test_explorer_sequential_equals_batched():
    c, env, robot = _setup(robot_visible=True)
    ex = c.Explorer(env, robot, 'cuda:0', gamma=0.9)
    ex.run_k_episodes(12, 'val')
    batched = dict(ex.last_stats)
    env.case_counter['val'] = 0
    stats = ex._run_sequential(12, 'val', False, False)
    ex._report(12, 'val', None, False, *stats)
    for key in ('success_rate', 'collision_rate', 'too_close', 'collision_cases', 'timeout_cases'):
        assert ex.last_stats[key] == batched[key], key
    assert abs(ex.last_stats['nav_time'] - batched['nav_time']) < 1e-12
    assert abs(ex.last_stats['total_reward'] - batched['total_reward']) < 1e-9


@pytest.mark.gpu
def test_debug_case_minus_one_and_square_crossing_through_gym_surface():
    """reset('test', -1): the reference's fixed 3-human layout (exact distance ties, crowd_sim.py:286-292); and
    env.test_sim = 'square_crossing' as test.py --square sets it (test.py:66-67)."""
    c, env, robot = _setup(robot_visible=False)
    g = load_golden('traj_debug_case.npz')
    e = episodes_of(g)[0]
    ob = env.reset('test', -1)
    assert env.human_num == 3 and len(ob) == 3
    assert np.array_equal(np.array([[h.px, h.py, h.gx, h.gy] for h in env.humans]), e['states'][0][1:, [0, 1, 4, 5]])
    for t in range(len(e['actions'])):
        action = robot.act(ob)
        assert (action.vx, action.vy) == tuple(e['actions'][t])
        ob, reward, done, info = env.step(action)
        assert reward == e['rewards'][t]
        assert np.array_equal(np.array([[h.px, h.py, h.vx, h.vy] for h in env.humans]), e['states'][t + 1][1:, :4])

    c, env, robot = _setup(robot_visible=True)
    env.test_sim = 'square_crossing'
    g = load_golden('traj_visible_h5_square.npz')
    e = episodes_of(g)[0]
    ob = env.reset('test', int(g['cases'][0]))
    got0 = np.array([[h.px, h.py, h.gx, h.gy] for h in env.humans])
    assert np.abs(got0 - e['states'][0][1:, [0, 1, 4, 5]]).max() <= 1e-12  # no trigonometry in this rule: exact
    assert np.array_equal(got0, e['states'][0][1:, [0, 1, 4, 5]])
    for t in range(len(e['actions'])):
        action = robot.act(ob)
        assert (action.vx, action.vy) == tuple(e['actions'][t])
        ob, reward, done, info = env.step(action)
        assert reward == e['rewards'][t] and done == bool(e['dones'][t])
    assert done


@pytest.mark.gpu
def test_get_human_times_vs_reference():
    """CrowdSim.get_human_times (crowd_sim.py:209-249): the centralised float32 rvo2 continuation after the robot has
    arrived — same arrival times, same number of simulated steps, same final positions as the unmodified reference
    (fixture: oracle/gen_golden_human_times.py)."""
    from conftest import load_golden
    import crowdnav_amd.compat as c
    g = load_golden('human_times.npz')
    cfg = c.default_env_config({('robot', 'visible'): 'true'})
    env = c.CrowdSim()
    env.configure(cfg)
    robot = c.Robot(cfg, 'robot')
    robot.set_policy(c.policy_factory['orca']())
    env.set_robot(robot)
    with pytest.raises(ValueError):
        env.reset('test', 0)
        env.get_human_times()  # episode is not done yet
    for k, case in enumerate(g['case'].tolist()):
        env.reset('test', case)
        # continue from the reference's own end-of-episode state (device scenarios differ from numpy's at 1e-12)
        env._eng.set_state(g['end_state'][k][None], np.array([g['end_time'][k]]))
        env._pull()
        env.human_times = list(g['times_before'][k])
        n0 = len(env.states)
        times = env.get_human_times()
        assert times == g['human_times'][k].tolist(), case
        assert len(env.states) - n0 == int(g['extra_steps'][k])
        got = np.array([[a.px, a.py] for a in [env.robot] + env.humans])
        assert np.array_equal(got, g['final_pos'][k]), case


def test_linear_policy_cpu():
    """policy_factory['linear'] (crowd_sim/envs/policy/linear.py): straight to the goal at v_pref."""
    import crowdnav_amd.compat as c
    pol = c.policy_factory['linear']()
    state = c.JointState(c.FullState(0.0, -4.0, 0.0, 0.0, 0.3, 3.0, 0.0, 1.0, np.pi / 2), [])
    a = pol.predict(state)
    assert abs(a.vx - 0.6) < 1e-12 and abs(a.vy - 0.8) < 1e-12 and not pol.trainable and pol.multiagent_training


@pytest.mark.gpu
def test_example_test_policy_script():
    """examples/test_policy.py = the reference's test.py: `--policy orca` over the 500 test cases gives the paper's
    ORCA row (success 0.43, collision 0.57, nav time 10.86: SURVEY.md §8(c)); one visible case ends with
    get_human_times."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location('test_policy', os.path.join(ROOT, 'examples', 'test_policy.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    stats = mod.run(mod.parser().parse_args(['--policy', 'orca']))
    assert round(stats['success_rate'], 2) == 0.43 and round(stats['collision_rate'], 2) == 0.57
    assert round(stats['nav_time'], 2) == 10.86
    one = mod.run(mod.parser().parse_args(['--policy', 'orca', '--visible', '--test-case', '1']))
    assert 'reach' in one['info'].lower() and len(one['human_times']) == 5 and all(t > 0 for t in one['human_times'])


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['persistent_orca.npz', 'persistent_orca_visible.npz'])
def test_persistent_orca_policy_with_random_attributes(name):
    """[env] randomize_attributes = true: the reference's ONE robot ORCA policy keeps the radii of the first episode's
    humans in its rvo2 simulator for every later episode (orca.py:95-110).  The batched Explorer shares that capture
    across its envs (cn_set_robot_sim) and so reproduces the unmodified reference case by case; so does the sequential
    loop on the gym surface."""
    from conftest import load_golden
    import crowdnav_amd.compat as c
    g = load_golden(name)
    k = int(g['k'])

    def make():
        cfg = c.default_env_config({('robot', 'visible'): 'true' if int(g['robot_visible']) else 'false',
                                    ('env', 'randomize_attributes'): 'true'})
        env = c.CrowdSim()
        env.configure(cfg)
        robot = c.Robot(cfg, 'robot')
        robot.set_policy(c.policy_factory['orca']())
        env.set_robot(robot)
        return c, env, robot

    c_, env, robot = make()
    ex = c_.Explorer(env, robot, 'cpu', gamma=0.9)
    ex.run_k_episodes(k, 'test')
    lb = ex.last_batch
    assert lb['outcome'] == g['outcome'].tolist() and lb['steps'] == g['steps'].tolist()
    assert np.abs(np.array(lb['discounted_return']) - g['returns']).max() <= 1e-9
    assert np.allclose(robot.policy._rsim[0], np.float32(g['first_radii'] + 0.01), rtol=0, atol=0)
    # the same through env.reset / robot.act / env.step
    c_, env, robot = make()
    ex = c_.Explorer(env, robot, 'cpu', gamma=0.9)
    robot.policy.set_phase('test')
    stats = ex._run_sequential(12, 'test', False, False)
    want = g['outcome'][:12]
    assert len(stats[0]) == int((want == 2).sum()) and len(stats[1]) == int((want == 3).sum())
    assert np.abs(np.array(stats[-1]) - g['returns'][:12]).max() <= 1e-9


def test_train_example_reads_the_reference_ini_files_cpu(tmp_path):
    """examples/train_sarl.py --env-config / --policy-config / --train-config take the reference's INI files as they
    are (crowd_nav/configs/*.config): same sections and keys."""
    import configparser
    import importlib.util
    import os
    from conftest import ROOT
    import crowdnav_amd.compat as c
    from crowdnav_amd.compat.sarl import default_policy_config
    spec = importlib.util.spec_from_file_location('train_sarl', os.path.join(ROOT, 'examples', 'train_sarl.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    train = configparser.RawConfigParser()
    train.read_dict({'trainer': dict(batch_size=64),
                     'imitation_learning': dict(il_episodes=7, il_policy='orca', il_epochs=3, il_learning_rate=0.02,
                                                safety_space=0.2),
                     'train': dict(rl_learning_rate=0.005, train_batches=9, train_episodes=11, sample_episodes=2,
                                   target_update_interval=4, evaluation_interval=6, capacity=500, epsilon_start=0.4,
                                   epsilon_end=0.2, epsilon_decay=8, checkpoint_interval=5)})
    for name, cfg in (('train.config', train), ('env.config', c.default_env_config()),
                      ('policy.config', default_policy_config())):
        with open(tmp_path / name, 'w') as f:
            cfg.write(f)
    args = mod.apply_train_config(mod.parser().parse_args([]), str(tmp_path / 'train.config'))
    assert (args.batch_size, args.il_episodes, args.il_epochs, args.train_batches, args.train_episodes) == (64, 7, 3, 9, 11)
    assert (args.il_learning_rate, args.safety_space, args.rl_learning_rate, args.epsilon_end) == (0.02, 0.2, 0.005, 0.2)
    assert (args.capacity, args.epsilon_decay, args.target_update_interval, args.sample_episodes) == (500, 8, 4, 2)
    env = c.CrowdSim()
    env.configure(mod.read_ini(str(tmp_path / 'env.config')))
    assert env.time_limit == 25 and env.human_num == 5 and env.case_size['test'] == 500
    policy = c.policy_factory['sarl']()
    policy.configure(mod.read_ini(str(tmp_path / 'policy.config')))
    assert policy.gamma == 0.9 and policy.multiagent_training and not policy.with_om
