#!/usr/bin/env python
"""Roll out a snapshotted policy (rll/rllab scripts/sim_policy.py:24-49 without the viewer):
    python scripts/sim_policy.py data/local/.../params.pkl --max_path_length 500 --n_paths 5
prints the undiscounted return of each path."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import joblib
    import numpy as np
    from rllab_amd.sampler.utils import rollout
    p = argparse.ArgumentParser()
    p.add_argument('file', type=str, help='path to the snapshot file')
    p.add_argument('--max_path_length', type=int, default=1000)
    p.add_argument('--n_paths', type=int, default=1)
    a = p.parse_args()
    data = joblib.load(a.file)
    policy, env = data['policy'], data['env']
    for _ in range(a.n_paths):
        path = rollout(env, policy, max_path_length=a.max_path_length)
        print("path length %d, return %.4f" % (len(path["rewards"]), float(np.sum(path["rewards"]))))


if __name__ == "__main__":
    main()
