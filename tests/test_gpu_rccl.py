"""The exchange family of the path through the REAL backend on a 1-GPU box: ``RLLAB_DIST_FORCE=1`` makes a
world of one rank count as distributed, so ``bench.py`` initialises ``torch.distributed`` with backend
``nccl`` (= RCCL) bound to the device and every all-reduce / all-gather / broadcast of an iteration is an
RCCL call (a one-rank collective is the identity, so results must not change).  What an 8-GPU run adds is
only peers; the API surface, stream semantics and the accounting fields of the JSON line are exercised here."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--n-envs", "512"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=env, cwd=ROOT, universal_newlines=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_iteration_over_rccl_single_rank():
    plain = _bench({})
    assert plain["ranks"] == 1 and plain["backend"] is None and plain["collectives_per_iter"] == 0
    forced = _bench({"RLLAB_DIST_FORCE": "1"})
    assert forced["ranks"] == 1 and forced["backend"].startswith("nccl")
    # per TRPO iteration: ONE all-gather of [statistics | normal equations], ONE of
    # [gradient | its loss sums], 10 Fisher-vector products, the two line-search candidates decided on the device
    assert 13 <= forced["collectives_per_iter"] <= 14.5, forced["collectives_per_iter"]
    # one rank: every per-rank column is this rank's, the spread is zero, no second reduction path to compare
    assert len(forced["per_rank_ms"]) == 1 and set(forced["per_rank_ms"][0]) == {"rank", "iteration", "sample", "process", "update"}
    assert all(v == 0 for v in forced["rank_skew_ms"].values()) and forced["other_sum_path"] is None
    assert abs(forced["per_rank_ms"][0]["iteration"] - forced["ms_per_step"]) < 1e-3
    assert forced["collective_bytes_per_iter"] < 1 << 20
    assert forced["collective_ms_per_iter"] is not None and 0 < forced["collective_ms_per_iter"] < 50
    assert forced["value"] > 0 and forced["config"]["n_envs_per_gpu"] == 512
    # the same with the in-stream peer all-reduce: the eleven sums on CG's critical path leave the host
    peer = _bench({"RLLAB_DIST_FORCE": "1", "RLLAB_PEER_ALLREDUCE": "1"})
    assert peer["peer_reductions_per_iter"] == 11 and forced["peer_reductions_per_iter"] == 0
    # (the gradient leaves its shared all-gather for the peer path; its loss sums keep one)
    assert abs(peer["collectives_per_iter"] - (forced["collectives_per_iter"] - 10)) < 1e-9
    assert peer["collectives_per_iter"] <= 5


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher around it: bench.py re-executes itself under torch.distributed.run
    (two ranks, rendezvous on 127.0.0.1), the ranks shard the envs and rank 0 prints ONE JSON line that reports what
    torch.distributed saw.  The GPU box has one device, so the ranks share it over gloo (the test backend); on an
    8-GPU node the same command runs one rank per GPU over RCCL."""
    env = dict(os.environ, RLLAB_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--n-envs", "256"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=ROOT,
                       universal_newlines=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["backend"] == "gloo" and d["scaling"] == "weak"
    assert d["config"]["samples_per_iteration"] == 2 * 256 * 500 and d["config"]["parallelism"] == "env-sharded dp2"
    assert 13 <= d["collectives_per_iter"] <= 14.5 and d["collective_ms_per_iter"] > 0
    assert d["value"] > 0 and "cpu_baseline" not in d
    # the line carries every rank's phase times and their spread, and -- measured by the same invocation -- the other
    # reduction path of the update's sums (the in-stream peer all-reduce), ready for the first 8-GPU lease
    assert [r["rank"] for r in d["per_rank_ms"]] == [0, 1] and all(r["iteration"] > 0 for r in d["per_rank_ms"])
    assert set(d["rank_skew_ms"]) == {"iteration", "sample", "process", "update"} and d["rank_skew_ms"]["iteration"] >= 0
    assert abs(max(r["iteration"] for r in d["per_rank_ms"]) - d["ms_per_step"]) < 1e-3      # `value` is the max over ranks
    o = d["other_sum_path"]
    assert d["update_sum_path"] == "backend" and o["update_sum_path"] == "peer"
    assert o["peer_reductions_per_iter"] == 11 and o["collectives_per_iter"] <= 5 and o["ms_per_step"] > 0


@pytest.mark.timeout(1500)
def test_bench_self_launches_eight_ranks():
    """The round-end scaling run's largest shape, `python bench.py --gpus 8`, rehearsed on the one-GPU box: eight ranks
    share the device over gloo at a reduced --n-envs (the 8-GPU node runs one rank per GPU over RCCL at 4096).  Every
    rank must get its own env index range, the exchange family must complete with eight peers, and the line must
    report eight times one rank's samples."""
    env = dict(os.environ, RLLAB_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                        "--n-envs", "64"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=ROOT,
                       universal_newlines=True, timeout=1400)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["ranks"] == 8 and d["backend"] == "gloo" and d["scaling"] == "weak"
    assert d["config"]["samples_per_iteration"] == 8 * 64 * 500 and d["config"]["parallelism"] == "env-sharded dp8"
    assert 13 <= d["collectives_per_iter"] <= 14.5 and d["collective_ms_per_iter"] > 0
    assert d["value"] > 0 and "cpu_baseline" not in d
    assert len(d["per_rank_ms"]) == 8 and d["other_sum_path"] is not None


def test_preflight_two_ranks_on_one_device():
    """tools/preflight_multigpu.py with a world of two (the box has one device: the ranks share it over gloo): devices
    and peer-access matrix reported, the in-stream peer all-reduce passes its staged check -- fine-grained hipIpc
    mailboxes mapped by the peer, a reduction of real-valued and of integer-valued rows bit for bit equal to the
    rank-ordered sum of the gathered rows and to the backend's all-reduce -- both paths timed over 100 calls, and the
    decision is the backend unless the peer path was asked for."""
    for asked in (False, True):
        env = dict(os.environ, RLLAB_DIST_BACKEND="gloo")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "RLLAB_PEER_ALLREDUCE"):
            env.pop(k, None)
        if asked:
            env["RLLAB_PEER_ALLREDUCE"] = "1"
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "preflight_multigpu.py"), "--gpus", "2"],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=ROOT, universal_newlines=True,
                           timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, p.stdout[-2000:]
        d = json.loads(lines[0])
        assert d["world"] == 2 and d["device_of_rank"] == [0, 0] and d["distinct_devices"] is False
        assert d["peer_access_between_ranks"] is True and d["peer_access_matrix"][0][0] is True
        assert d["peer_check"]["passed"] is True, d["peer_check"]
        assert d["backend_allreduce_us"] > 0 and d["peer_allreduce_us"] > 0 and len(d["per_rank_latency_us"]) == 2
        assert d["decision"] == ("peer" if asked else "backend") and d["peer_requested"] is asked


def test_bench_runs_the_preflight_before_warmup():
    """`bench.py --gpus 2` carries the pre-flight record (taken in child processes) in its JSON line, and a requested
    peer path that passed it is the path the update's sums took."""
    env = dict(os.environ, RLLAB_DIST_BACKEND="gloo", RLLAB_PEER_ALLREDUCE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--n-envs", "256"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=ROOT,
                       universal_newlines=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    pf = d["preflight"]
    assert pf["isolated"] is True and pf["world"] == 2 and pf["peer_check"]["passed"] is True
    assert pf["decision"] == "peer" and d["update_sum_path"] == "peer" and d["peer_reductions_per_iter"] == 11
