"""Build libodbox.so (the HIP kernels + C-ABI) in-tree for gfx950.

    python -m object_detection_amd.build

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the numerical contract
(no FMA contraction: the reference rounds every fp32 op separately)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'libodbox.so')
OBJ = os.path.join(HERE, 'build')
COMMON = ['--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-fPIC', '-std=c++17',
          '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC]
# translation unit -> extra flags (see the header comment of each file)
UNITS = {
    # the matcher's degenerate-box guards are NaN-aware comparisons (!(x >= 0)): NaNs are honoured in this unit
    # (ODBOX_TRAIN_FLAGS: per-unit extra flags of a tuning build, e.g. the A/B that decided this)
    'odbox_train.hip': os.environ.get('ODBOX_TRAIN_FLAGS', '').split(),
    'odbox_infer.hip': ['-fno-honor-nans'],
    'odbox_eval.hip': [],
}
HEADERS = [os.path.join(CSRC, 'odbox_common.h'), os.path.join(ROOT, 'include', 'odbox.h')]


STAMP = os.path.join(OBJ, 'flags.txt')


def tuning_flags():
    # ODBOX_TUNING_HOOKS=1: the library reads the ODBOX_SCAN_* / ODBOX_NMS_DBG experiment knobs (tools/exp_*.sh)
    hooks = ['-DODBOX_TUNING_HOOKS'] if os.environ.get('ODBOX_TUNING_HOOKS') == '1' else []
    return (hooks + os.environ.get('ODBOX_EXTRA_HIPCC_FLAGS', '').split()      # tuning builds only, e.g. -DODBOX_ME_TILE_ENC=512
            + ['@train:' + f for f in os.environ.get('ODBOX_TRAIN_FLAGS', '').split()])


def needs_build():
    if not os.path.exists(OUT):
        return True
    # a library built with other flags than the ones asked for now is stale, whatever its age
    built_with = open(STAMP).read() if os.path.exists(STAMP) else ''
    if built_with != ' '.join(tuning_flags()):
        return True
    deps = HEADERS + [os.path.join(CSRC, u) for u in UNITS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps)


def check_m0(hipcc, hooks, verbose=False):
    """odbox_infer.hip sets M0 from inline asm for its LDS-DMA loads (s_mov_b32 m0 + global_load_lds in
    ONE asm statement).  M0 is a reserved register that hipcc refuses in a clobber list, so the
    assumption "the compiler keeps nothing of its own in M0 across those statements" is verified on
    the generated code instead: in the device assembly of that unit every instruction that names m0
    must be one of those s_mov_b32, directly followed by its s_nop and its global_load_lds."""
    unit = 'odbox_infer.hip'
    asm = os.path.join(OBJ, 'odbox_infer.s')
    cmd = [hipcc] + COMMON + UNITS[unit] + hooks + ['-S', '--cuda-device-only', '-Wno-unused-command-line-argument',
                                                    os.path.join(CSRC, unit), '-o', asm]
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    # only the kernels that issue the LDS-DMA (score_scan_kernel<...>) matter: elsewhere the compiler is free to
    # use M0 (e.g. as the lane select of a v_readlane in the NMS kernel)
    lines, inside = [], False
    for raw in open(asm):
        l = raw.split(';')[0].strip()
        if l.endswith(':') and not l.startswith('.'):
            inside = 'score_scan_kernel' in l
        elif l.startswith('.Lfunc_end'):
            inside = False
        elif inside and l and not l.startswith(('.', '//')):
            lines.append(l)
    sets = 0
    for i, l in enumerate(lines):
        if 'm0' not in l.replace(',', ' ').split():
            continue
        ok = (l.startswith('s_mov_b32 m0,') and i + 2 < len(lines) and lines[i + 1].startswith('s_nop')
              and lines[i + 2].startswith('global_load_lds_dword'))
        if not ok:
            raise RuntimeError('%s: an instruction outside the LDS-DMA asm statements uses m0: %r (see dma16 / dma4)' % (asm, l))
        sets += 1
    if sets == 0:
        raise RuntimeError('%s: no LDS-DMA found -- the check is looking at the wrong file' % asm)
    return sets


def check_tail_vgprs(limit=96):
    """The fused pass-1 block of decode+NMS (tier_tail_kernel: 1 024 threads, four waves per SIMD) has to become resident
    on a CU whose SIMDs already hold one or two waves of ANOTHER batch's score scan (56 VGPRs each), or two batches in
    flight (bndbox.DecodeNmsPipeline) stop overlapping: 4 x 96 + 2 x 56 <= 512.  __launch_bounds__(TT_THREADS, 5) asks
    for that; this reads what the compiler really allocated from the assembly check_m0 has just written (measured:
    118 VGPRs -> two batches in flight 195 us per batch instead of 180)."""
    asm = os.path.join(OBJ, 'odbox_infer.s')
    for raw in open(asm):
        l = raw.strip()
        if l.startswith('.set') and 'tier_tail_kernel' in l and '.num_vgpr,' in l:
            n = int(l.rsplit(',', 1)[1])
            if n > limit:
                raise RuntimeError('tier_tail_kernel takes %d VGPRs (limit %d): it no longer fits beside a score scan' % (n, limit))
            return n
    raise RuntimeError('%s: tier_tail_kernel not found -- the check is looking at the wrong file' % asm)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get('HIPCC', 'hipcc')
    hooks = [f for f in tuning_flags() if not f.startswith('@train:')]
    os.makedirs(OBJ, exist_ok=True)
    objs = []
    for unit, extra in UNITS.items():
        obj = os.path.join(OBJ, unit.replace('.hip', '.o'))
        cmd = [hipcc] + COMMON + extra + hooks + ['-c', os.path.join(CSRC, unit), '-o', obj]
        if verbose:
            print(' '.join(cmd))
        subprocess.run(cmd, check=True)
        objs.append(obj)
    check_m0(hipcc, hooks, verbose)
    check_tail_vgprs()
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    with open(STAMP, 'w') as f:
        f.write(' '.join(tuning_flags()))
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv, verbose=True)
    print(OUT)
