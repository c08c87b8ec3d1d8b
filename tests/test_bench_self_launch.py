"""bench.py started as `python bench.py --gpus N` WITHOUT a launcher (no WORLD_SIZE in the
environment) must start its own N ranks (torch.distributed.run on 127.0.0.1) instead of exiting --
the shape of the driver's N = 1 command, applied to N > 1.

CPU: the launch path alone (`--rendezvous-only`: process group + one all_reduce of ones, over gloo).
GPU: the whole bench with two self-started ranks sharing the box's one GPU over gloo
(GRPG_BENCH_BACKEND=gloo is the documented single-GPU aid; the RCCL path needs two devices).
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE",
                        "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
    env.update(extra)
    return env


def _json_lines(text):
    out = []
    for ln in text.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                out.append(json.loads(ln))
            except ValueError:
                pass
    return out


def test_self_launch_forms_the_group_over_gloo():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                       env=_clean_env(GRPG_BENCH_BACKEND="gloo"), cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout          # rank 0 alone prints
    assert lines[0]["n_gpus"] == 2 and lines[0]["rccl_ranks_seen"] == 2
    assert "starting 2 ranks" in p.stderr


def test_launcher_mismatch_is_an_error_not_a_hang():
    # WORLD_SIZE set by someone else and different from --gpus: refuse, loudly
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--rendezvous-only"],
                       env=_clean_env(WORLD_SIZE="2", RANK="0"), cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


def test_rccl_with_too_few_devices_fails_fast_with_the_remedy():
    """`--gpus 2` over RCCL (the default backend) on a node with fewer than two devices: refused before any rank
    is started, with a message that names the device count and the way out -- not N ranks dying inside RCCL's
    init.  On a node that HAS two devices the same command simply forms the group."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices present: the command succeeds (covered by the driver's multi-GPU run)")
    env = _clean_env()
    env.pop("GRPG_BENCH_BACKEND", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0
    assert "needs 2 ROCm devices" in p.stderr and "GRPG_BENCH_BACKEND=gloo" in p.stderr
    assert "starting 2 ranks" not in p.stderr


@pytest.mark.gpu
def test_self_launched_two_ranks_render_on_one_gpu():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6",
                        "--warmup", "2", "--gaussians", "200000", "--no-train", "--no-strong",
                        "--no-delivery", "--no-cpu-baseline", "--no-secondary"],
                       env=_clean_env(GRPG_BENCH_BACKEND="gloo"), cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in _json_lines(p.stdout) if "metric" in ln]
    assert len(lines) == 1
    ln = lines[0]
    assert ln["n_gpus"] == 2 and ln["rccl_ranks_seen"] == 2 and ln["steps"] == 6
    assert ln["value"] > 0 and ln["scaling"] == "weak"
