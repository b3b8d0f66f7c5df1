#!/usr/bin/env python3
"""fps of the batched evaluation driver (trainer/batched_evaluation.py) at 4 096 lock-step episodes: host-stepped loop
(one host round trip per step: root states up, plans down, numpy env step) vs the device-resident loop (mp_env_step on the
planner's root-state buffer).  Headline MCTS configuration on the highway-shaped table; also OPD and a VI agent.

    python tools/eval_fps.py [episodes]      -> profiles/r03_batched_eval_fps.txt (through gpurun)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from rl_agents_amd.agents.common.factory import agent_factory
    from rl_agents_amd.envs import FiniteMDPEnv, generators
    from rl_agents_amd.trainer.batched_evaluation import BatchedEvaluation
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    cfg = dict(generators.highway_shaped(10, 10, 100, seed=0), max_steps=30)
    non_term = np.flatnonzero(~np.asarray(cfg["terminal"]))
    starts = np.random.Generator(np.random.PCG64(1)).choice(non_term, size=n).astype(np.int32)
    agents = [("MCTSAgent budget 1000 (33 x 30)", dict(__class__="<class 'rl_agents_amd.agents.tree_search.mcts.MCTSAgent'>",
                                                    budget=1000, horizon=30, episodes=33)),
              ("DeterministicPlannerAgent budget 500", dict(__class__="<class 'rl_agents_amd.agents.tree_search.deterministic.DeterministicPlannerAgent'>",
                                                          budget=500, gamma=0.8)),
              ("ValueIterationAgent", dict(__class__="<class 'rl_agents_amd.agents.dynamic_programming.value_iteration.ValueIterationAgent'>",
                                           gamma=0.95, iterations=200))]
    print("batched evaluation, {} lock-step episodes, highway-shaped S = 10 000, |A| = 5, episode cap 30 steps".format(n))
    for name, acfg in agents:
        row = []
        for resident in (False, True):
            env = FiniteMDPEnv(dict(cfg))
            env.reset()
            agent = agent_factory(env, dict(acfg))
            ev = BatchedEvaluation(env, agent, num_episodes=n, sim_seed=0, max_steps=30, device_resident=resident)
            ev.run(initial_states=starts)                       # warm-up (uploads, tables)
            t0 = time.perf_counter()
            out = ev.run(initial_states=starts)
            wall = time.perf_counter() - t0
            row.append((out["fps"], wall, float(out["lengths"].mean()), float(out["returns"].mean())))
        (f0, w0, l0, r0), (f1, w1, l1, r1) = row
        assert l0 == l1 and r0 == r1, "the two loops must give the same episodes"
        print("{:42s} host-stepped {:12.0f} env steps/s ({:7.2f} ms)   device-resident {:12.0f} env steps/s ({:7.2f} ms)   x{:.2f}   "
              "mean length {:.1f}".format(name, f0, 1e3 * w0, f1, 1e3 * w1, f1 / f0, l0))


if __name__ == "__main__":
    main()
