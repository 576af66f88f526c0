#!/usr/bin/env python3
"""Known-answer fixture for the configs[4] EXTENSION (scripted bodies, curriculum levels, ended / truncated).

The reference has nothing to generate this from (no obstacle entity, one world per env object): the fixture is produced
by the build's own oracle (oracle/uavx_oracle.c, uavo_*_x) and pins the DEFINITION of include/uavx.h in data form, so that
a later change to either restatement shows up as a diff against committed numbers.  It is NOT a reference vector; parity of
the extension stays "unpinned by the reference".

    python tests/golden/make_ext_golden.py        -> tests/golden/ext_bodies_levels.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402

LEVELS = [dict(x_size=14.0, y_size=12.0, collider_radius=0.4, d_sense=6.0, n_active=1, b_active=2),
          dict(x_size=20.0, y_size=20.0, collider_radius=0.8, d_sense=9.0, n_active=3, b_active=5)]
KW = dict(num_agents=3, num_bodies=5, body_speed=3.0, body_period=4, body_seed=11, x_size=20.0, y_size=20.0, d_sense=9.0)
E, T, SEED, OFFSET, CAP = 6, 40, 5, 17, 13


def run():
    orc = oracle.OracleMulti(num_envs=E, **KW)
    orc.set_curriculum(LEVELS, lo=0, hi=1)
    orc.reset_philox(SEED, env_offset=OFFSET)
    rng = np.random.default_rng(1)
    out = dict(init_loc=orc.loc.copy(), init_tgt=orc.tgt.copy(), init_body=orc.body.copy(), init_level=orc.level.copy(),
               init_flags=orc.flags.copy())
    rec = {k: [] for k in ("actions", "obs", "rew", "done", "reset_mask", "ended", "truncated", "loc", "vel", "body", "level",
                           "flags", "counters")}
    for t in range(T):
        a = rng.uniform(-1, 1, size=(E, KW["num_agents"], 2)).astype(np.float32)
        obs, rew, done, rm, en, tr = orc.step_ex(a, action_mode=1, reset_policy=2, step_cap=CAP, track_returns=True, seed=SEED,
                                                 env_offset=OFFSET, evaluate=True, with_end=True)
        for k, v in (("actions", a), ("obs", obs), ("rew", rew), ("done", done), ("reset_mask", rm), ("ended", en),
                     ("truncated", tr), ("loc", orc.loc), ("vel", orc.vel), ("body", orc.body), ("level", orc.level),
                     ("flags", orc.flags), ("counters", orc.counters)):
            rec[k].append(np.array(v, copy=True))
    out.update({k: np.stack(v) for k, v in rec.items()})
    return out


META = dict(kind="extension", source="oracle/uavx_oracle.c (uavo_*_x): NOT a reference vector, see the module docstring",
            kw=KW, levels=LEVELS, envs=E, steps=T, seed=SEED, env_offset=OFFSET, step_cap=CAP)

if __name__ == "__main__":
    import json
    oracle.build()
    np.savez_compressed(os.path.join(HERE, "ext_bodies_levels.npz"), meta=np.array(json.dumps(META)), **run())
    print("wrote ext_bodies_levels.npz")
