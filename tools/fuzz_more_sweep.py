"""One-off extension of tests/test_gpu_fuzz.py::test_sweep_random_shapes_and_placements to seeds 64 .. 699 (636 more shapes; last run: 0 failures)."""
import sys, os
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import importlib.util
spec = importlib.util.spec_from_file_location('fz', 'tests/test_gpu_fuzz.py')
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
bad = 0
for seed in range(64, 700):
    try:
        fz.test_sweep_random_shapes_and_placements(seed)
    except AssertionError as e:
        bad += 1
        print('seed', seed, 'FAILED', str(e)[:200], flush=True)
    if seed % 100 == 0:
        print('seed', seed, 'bad so far', bad, flush=True)
print('done, failures:', bad)
