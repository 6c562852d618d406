"""CPU: `python bench.py --gpus N` must start N ranks by itself (the driver's scaling run calls it exactly like that), relay
ONE result line from rank 0 and fail loudly when a rank fails or the world size disagrees with --gpus (missing #1).
The rehearsal uses gloo and a stub step (`--dry-run`): the launcher, the rendezvous, the barriers and the MAX-over-ranks timing are
the code the GPU run uses."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_gpus_flag_starts_that_many_ranks():
    r = _run(['--gpus', '2', '--dry-run', '--steps', '3', '--warmup', '1'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                      # exactly one line on stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['world_size_observed'] == 2 and out['rank_sum'] == 1.0     # ranks 0 and 1 both took part
    assert out['steps'] == 3 and out['warmup'] == 1 and out['dry_run'] is True and out['value'] is None


def test_single_rank_needs_no_launcher():
    r = _run(['--dry-run', '--steps', '2', '--warmup', '0'])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip())['n_gpus'] == 1


def test_world_size_mismatch_is_an_error():
    r = _run(['--gpus', '4', '--dry-run'], {'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and '--gpus 4 but WORLD_SIZE=2' in r.stderr and r.stdout.strip() == ''


def test_failing_rank_fails_the_launch():
    # without a GPU every rank of the real (non-dry) run stops at its `needs a GPU` assertion: the parent must report failure
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip('needs a host without GPUs')
    r = _run(['--gpus', '2', '--steps', '1', '--warmup', '0'])
    assert r.returncode != 0 and r.stdout.strip() == ''
