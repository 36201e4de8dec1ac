"""Build libodbox.so (the HIP kernels + C-ABI) in-tree for gfx950.

    python -m object_detection_amd.build

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the numerical contract
(no FMA contraction: the reference rounds every fp32 op separately)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, 'csrc', 'odbox.hip')
OUT = os.path.join(HERE, 'libodbox.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-fPIC', '-shared', '-std=c++17',
         '-I' + os.path.join(ROOT, 'include')]


def needs_build():
    if not os.path.exists(OUT):
        return True
    deps = [SRC, os.path.join(ROOT, 'include', 'odbox.h')]
    return any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get('HIPCC', 'hipcc')
    cmd = [hipcc] + FLAGS + ['-o', OUT, SRC]
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv, verbose=True)
    print(OUT)
