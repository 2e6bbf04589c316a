"""rllab_amd -- MI355X-native batched rollout + policy-gradient engine behind the
rllab Env / Policy / Baseline / Sampler / optimizer API."""
__version__ = "0.1.0"
