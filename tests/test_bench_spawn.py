"""bench.py's launch contract, on CPU: `python bench.py --gpus N` invoked plainly starts its own N
ranks (children, through torch.distributed.run), relays rank 0's JSON line and the children's exit
status, and refuses -- non-zero, no fallback to fewer ranks -- when fewer than N GPUs are visible.
`--selftest-child` swaps the GPU work for one MAX all-reduce over gloo, so this runs without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'ODBOX_DIST_BACKEND')}
    e.update(env)
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=300)


def test_plain_invocation_starts_its_own_ranks_and_relays_rank0():
    r = _run(['--gpus', '2', '--selftest-child'], ODBOX_DIST_BACKEND='gloo')
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1                                   # rank 0 only
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['max_rank'] == 1.0     # both ranks joined the collective


def test_a_failing_rank_fails_the_parent():
    r = _run(['--gpus', '2', '--selftest-child', 'fail'], ODBOX_DIST_BACKEND='gloo')
    assert r.returncode != 0


def test_too_few_gpus_is_an_error_not_a_smaller_run():
    import torch
    if torch.cuda.device_count() >= 16:
        return
    r = _run(['--gpus', '16', '--selftest-child'])
    assert r.returncode != 0 and 'refusing to run fewer ranks' in r.stderr and '{' not in r.stdout


def test_world_size_that_contradicts_gpus_is_an_error():
    r = _run(['--gpus', '4', '--selftest-child'], WORLD_SIZE='2', RANK='0')
    assert r.returncode != 0 and 'WORLD_SIZE' in r.stderr


def test_single_process_needs_no_launcher():
    r = _run(['--gpus', '1', '--selftest-child'])
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])['n_gpus'] == 1
