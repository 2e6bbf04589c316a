#!/usr/bin/env python
"""Entry script (subset of rll/rllab scripts/run_experiment_lite.py:21-139 the local mode needs):
    python scripts/run_experiment_lite.py --resume_from data/local/.../params.pkl [--seed N] [--snapshot_mode last]
resumes a snapshotted algorithm (``algo.train()`` continues at ``algo.current_itr``)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv):
    from rllab_amd.misc.instrument import run_experiment_lite
    p = argparse.ArgumentParser()
    p.add_argument('--n_parallel', type=int, default=1)
    p.add_argument('--exp_name', type=str, default=None)
    p.add_argument('--log_dir', type=str, default=None)
    p.add_argument('--snapshot_mode', type=str, default='all')
    p.add_argument('--snapshot_gap', type=int, default=1)
    p.add_argument('--seed', type=int, default=None)
    p.add_argument('--resume_from', type=str, required=True)
    a = p.parse_args(argv[1:])
    run_experiment_lite(resume_from=a.resume_from, exp_name=a.exp_name, log_dir=a.log_dir, n_parallel=a.n_parallel,
                        snapshot_mode=a.snapshot_mode, snapshot_gap=a.snapshot_gap, seed=a.seed)


if __name__ == "__main__":
    main(sys.argv)
