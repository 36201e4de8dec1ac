"""The C-ABI used from a plain host program (tests/c_abi/odbox_c_check.cpp: hipMalloc'd buffers, a
hipStream_t, no Python, no torch) -- what a C++ data loader or serving loop links against.  The
binary is built in-tree by __graft_entry__.build(); it checks both hot paths against the C oracle."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, 'c_abi', 'odbox_c_check')


def test_plain_c_host_program_matches_the_oracle():
    # under -m gpu a missing binary is a failure, not a skip: this is the only non-Python check of the C-ABI
    assert os.path.exists(BIN), 'tests/c_abi/odbox_c_check is not built (python __graft_entry__.py builds it)'
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=120)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith('PASS')
    assert '0 index/state/one-hot mismatches, 0 offsets' in r.stdout
    assert '0 keep/label/score/count mismatches, 0 boxes' in r.stdout
    assert 'decode_nms_mixed (bf16 scores): 0 keep/label/score/count mismatches, 0 boxes' in r.stdout
