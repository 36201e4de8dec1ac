"""Soak run of the seeded fuzz tests of tests/test_gpu_fuzz.py beyond the seeds pytest collects: every case is the HIP
path against the C oracle, bit for bit (fp32 tolerance for the loss, as written in the test).

    python tools/fuzz_more.py NAME LO HI      NAME: match_encode | match_loss | decode_nms | sweep | fused
"""
import importlib.util
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
spec = importlib.util.spec_from_file_location('fz', 'tests/test_gpu_fuzz.py')
fz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fz)

CASES = {
    'match_encode': fz.test_match_encode_random_shapes_and_placements,
    'match_loss': fz.test_match_loss_random_shapes,
    'decode_nms': fz.test_decode_nms_random_shapes,
    'sweep': fz.test_sweep_random_shapes_and_placements,
    'fused': fz.test_fused_pass_one_block_random_shapes,
}
name, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
case = CASES[name]
bad = 0
for seed in range(lo, hi):
    try:
        case(seed)
    except AssertionError as e:
        bad += 1
        print(name, 'seed', seed, 'FAILED', str(e)[:300], flush=True)
    if seed % 50 == 0:
        print(name, 'seed', seed, 'bad so far', bad, flush=True)
print(name, 'seeds', lo, '..', hi - 1, 'done, failures:', bad, flush=True)
sys.exit(1 if bad else 0)
