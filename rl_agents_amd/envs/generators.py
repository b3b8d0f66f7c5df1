"""Seeded synthetic finite-MDP tables for the BASELINE.json configurations.

``gymnasium``, ``finite_mdp`` and ``highway_env`` are absent from this image, so the
configurations that name them are realised as *shaped* tables.  Every report
that uses them says "synthetic, highway-/intersection-shaped".  All generators return a
finite-MDP config dict in the reference's wire format
(scripts/configs/FiniteMDPEnv/large/env_1.json: mode / transition / reward / terminal).
"""
import numpy as np


def gridworld(width=10, height=10, goal=(7, 7), radius2=25.0, n_actions=4):
    """C1: deterministic GridWorld, S = width*height, A = 4 (+x, -x, +y, -y clamped at walls).

    Reward shape borrowed from utils/envs/gridenv.py:52-55 (radial bump around a goal), clipped
    to [0, 1] so that OPD accepts it (deterministic.py:46-47).  No terminal states.
    """
    s = width * height
    xs, ys = np.meshgrid(np.arange(width), np.arange(height), indexing="ij")
    idx = (xs * height + ys)
    moves = [(1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, -1), (1, -1), (-1, 1)][:n_actions]
    transition = np.zeros((s, n_actions), dtype=np.int64)
    for a, (dx, dy) in enumerate(moves):
        nx = np.clip(xs + dx, 0, width - 1)
        ny = np.clip(ys + dy, 0, height - 1)
        transition[idx.ravel(), a] = (nx * height + ny).ravel()
    cell_reward = np.clip(1.0 - ((xs - goal[0]) ** 2 + (ys - goal[1]) ** 2) / radius2, 0.0, 1.0)
    # reward for acting in a cell = bump value of the cell that is reached
    reward = cell_reward.ravel()[transition]
    return dict(mode="deterministic", transition=transition, reward=reward,
                terminal=np.zeros(s, dtype=bool))


def highway_shaped(n_speeds=10, n_lanes=10, n_times=100, collision_rate=0.05, seed=0):
    """C2 / headline: highway-shaped deterministic table over a (speed, lane, time) grid.

    Mirrors the *layout* of highway_env's time-to-collision MDP [from memory, package absent]:
    A = 5 = (LANE_LEFT, IDLE, LANE_RIGHT, FASTER, SLOWER); time advances every step and
    saturates at the last slice; collision cells ~ Bernoulli(collision_rate); terminal =
    collision or last time slice; reward in [0, 1] favours free cells, right lanes, high speed.
    Default (10, 10, 100) -> S = 10 000.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    v, l, t = np.meshgrid(np.arange(n_speeds), np.arange(n_lanes), np.arange(n_times), indexing="ij")
    shape = (n_speeds, n_lanes, n_times)
    s = n_speeds * n_lanes * n_times

    def flat(vv, ll, tt):
        return (vv * n_lanes + ll) * n_times + tt

    tn = np.minimum(t + 1, n_times - 1)
    nxt = [
        flat(v, np.maximum(l - 1, 0), tn),                 # LANE_LEFT
        flat(v, l, tn),                                    # IDLE
        flat(v, np.minimum(l + 1, n_lanes - 1), tn),       # LANE_RIGHT
        flat(np.minimum(v + 1, n_speeds - 1), l, tn),      # FASTER
        flat(np.maximum(v - 1, 0), l, tn),                 # SLOWER
    ]
    transition = np.stack([n.ravel() for n in nxt], axis=1).astype(np.int64)
    collision = rng.random(shape) < collision_rate
    cell_reward = np.clip(0.5 * (~collision) + 0.1 * l / max(n_lanes - 1, 1)
                          + 0.4 * v / max(n_speeds - 1, 1), 0.0, 1.0)
    reward = np.repeat(cell_reward.reshape(s, 1), 5, axis=1)
    terminal = (collision | (t == n_times - 1)).ravel()
    return dict(mode="deterministic", transition=transition, reward=np.ascontiguousarray(reward),
                terminal=terminal, original_shape=shape)


def rewire(config, fraction=0.1, seed=1):
    """Second model for robust VI (C5): the same table with a seeded fraction of transitions rewired."""
    rng = np.random.Generator(np.random.PCG64(seed))
    transition = np.array(config["transition"], dtype=np.int64, copy=True)
    s, a = transition.shape
    mask = rng.random((s, a)) < fraction
    transition[mask] = rng.integers(0, s, size=int(mask.sum()))
    out = dict(config)
    out["transition"] = transition
    return out


def random_deterministic(n_states, n_actions, seed=0, terminal_rate=0.0):
    """Garnet-like deterministic table: uniform random successors, rewards U[0,1)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    transition = rng.integers(0, n_states, size=(n_states, n_actions), dtype=np.int64)
    reward = rng.random((n_states, n_actions))
    terminal = rng.random(n_states) < terminal_rate
    return dict(mode="deterministic", transition=transition, reward=reward, terminal=terminal)


def random_stochastic(n_states, n_actions, seed=0, terminal_rate=0.0, concentration=0.3):
    """Dense row-stochastic ``transition[S, A, S]`` (Dirichlet rows), rewards U[0,1)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    w = rng.gamma(concentration, size=(n_states, n_actions, n_states))
    w /= w.sum(axis=-1, keepdims=True)
    reward = rng.random((n_states, n_actions))
    terminal = rng.random(n_states) < terminal_rate
    return dict(mode="stochastic", transition=w, reward=reward, terminal=terminal)


def random_sparse(n_states, n_actions, branching=2, seed=0, terminal_rate=0.0):
    """Garnet table in the reference's sparse format: ``transition[S,A,B]`` probs, ``next[S,A,B]`` ids."""
    rng = np.random.Generator(np.random.PCG64(seed))
    nxt = rng.integers(0, n_states, size=(n_states, n_actions, branching), dtype=np.int64)
    p = rng.random((n_states, n_actions, branching)) + 0.05
    p /= p.sum(axis=-1, keepdims=True)
    reward = rng.random((n_states, n_actions))
    terminal = rng.random(n_states) < terminal_rate
    return dict(mode="sparse", transition=p, next=nxt, reward=reward, terminal=terminal)


def highway_available(config):
    """Action availability of a highway-shaped table in highway-env's style [from memory, package absent]: no
    LANE_LEFT in the leftmost lane, no LANE_RIGHT in the rightmost, no FASTER at the top speed, no SLOWER at the lowest;
    IDLE is always available.  -> bool [S, 5]."""
    from rl_agents_amd.device_model import grid_available
    return grid_available(config["original_shape"])


def random_available(n_states, n_actions, seed=0, rate=0.3):
    """Seeded availability table: each (state, action) unavailable with probability ``rate``, at least one action
    available in every state."""
    rng = np.random.Generator(np.random.PCG64(seed))
    avail = rng.random((n_states, n_actions)) >= rate
    avail[np.arange(n_states), rng.integers(0, n_actions, size=n_states)] = True
    return avail
