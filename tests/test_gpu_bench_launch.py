"""gpu: `python3 bench.py --gpus N` must start by itself (the driver starts the bench plain: no launcher, no
WORLD_SIZE).  On a one-GPU box the N ranks share cuda:0 and the halos travel over gloo (RCCL refuses two ranks on
one device) -- the numbers mean nothing, the launch path, the rank bookkeeping and the ONE JSON line are what is
checked.  Also the launcher form the task contract names (torch.distributed.run) keeps working."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ)
    env.update(AKMI_SHARE_GPU="1", AKMI_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _one_json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_gpus_2_starts_its_own_ranks():
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--nx", "64", "--steps", "3", "--warmup", "1"],
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    r = _one_json_line(p.stdout)
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1 and r["scaling"] == "weak"
    assert r["value"] > 0 and r["unit"] == "Mcell-updates/s"
    assert "128x64x64" in r["config"]["workload"]          # 2 x 1 x 1 blocks of 64^3: per-GPU work fixed
    assert "cpu_baseline" not in r                          # N = 1 only


def test_bench_gpus_8_is_the_2x2x2_layout():
    """the launch the driver makes on an 8-GPU node, shrunk to 32^3 per rank on a shared GPU: eight ranks, every one of
    the 26 neighbours of a block on another rank, one JSON line.  (RCCL refuses eight ranks on one device, so the
    messages travel over gloo and the C++/RCCL host, which needs one GPU per rank, says so in config.host_check.)"""
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--nx", "32", "--steps", "2", "--warmup", "1"],
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    r = _one_json_line(p.stdout)
    assert r["n_gpus"] == 8 and r["steps"] == 2 and r["scaling"] == "weak" and r["value"] > 0
    assert "64x64x64" in r["config"]["workload"] and "2x2x2" in r["config"]["workload"]
    assert "host_check" in r["config"] and "roofline" in r
    # the exchange taken apart (whichever host wrote the line): an 8-GPU run can be attributed from the record
    halo = r["roofline"]["halo"]
    for k in ("pack_ms", "exposed_wait_ms", "unpack_ms", "dt_reduce_ms", "bytes_sent_per_stage", "peers"):
        assert k in halo, (k, halo)
    assert halo["peers"] == 7 and halo["bytes_sent_per_stage"] > 0 and halo["posts_per_stage"] == 2.0   # U and B, 7 peers each
    assert halo["pack_ms"] > 0 and halo["unpack_ms"] > 0


def test_bench_under_torch_distributed_run():
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29631", "bench.py", "--gpus", "2", "--nx", "64",
                        "--steps", "2", "--warmup", "1"],
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    r = _one_json_line(p.stdout)
    assert r["n_gpus"] == 2 and r["steps"] == 2


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = _env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--nx", "64", "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr
