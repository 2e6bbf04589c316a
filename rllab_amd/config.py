"""Project paths (the subset of rllab/config.py the local experiment runner reads)."""
import os

PROJECT_PATH = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
LOG_DIR = os.environ.get("RLLAB_LOG_DIR", os.path.join(PROJECT_PATH, "data"))
