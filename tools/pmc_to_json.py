"""Turn the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh into
profiles/pmc_traffic.json: HBM bytes per launch of the dominant kernels.

    python tools/pmc_to_json.py gpurun_out/round/pmc

Layout: <root>/{stream,auto,resident}/<COUNTER>/... = tools/pmc_headline.py (configs[1] match+encode: five
output sets in rotation under the 'stream' / default policy, one set rewritten in place);
<root>/rest/<COUNTER>/... = bench.py --no-configs (loss and decode+NMS kernels).

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM): both counters are in KiB; on gfx950
FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane loads
and LDS-DMA alike) -> doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
Keys of the old file that these passes do not measure (e.g. match_encode_cfg3_valu_busy) are kept."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/round/pmc'
REST = (('match_loss_kernel_fwd', 'match_loss_kernel<float, false'),
        ('match_loss_kernel_fwd_bwd', 'match_loss_kernel<float, true'),
        ('match_loss_kernel_f16_fwd', 'match_loss_kernelIDF16_Lb0'),     # rocprofv3 leaves _Float16
        ('match_loss_kernel_f16_bwd', 'match_loss_kernelIDF16_Lb1'),     # instantiations mangled
        ('score_scan_kernel', 'score_scan_kernel'), ('nms_kernel', 'nms_kernel'),
        ('finalize_kernel', 'finalize_kernel'), ('cand_partition_kernel', 'cand_partition_kernel'))
HEAD = {'stream': 'match_encode_kernel_rotating', 'auto': 'match_encode_kernel_rotating_auto_policy',
        'resident': 'match_encode_kernel'}
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ('stream', 'auto', 'resident', 'rest'):
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        # newest run only: gpurun merges results into gpurun_out/ and never deletes an older run's files
        for f in sorted(glob.glob(os.path.join(root, sub, counter, '*', '*counter_collection.csv')), key=os.path.getmtime)[-1:]:
            for r in csv.DictReader(open(f)):
                name = r['Kernel_Name']
                if sub == 'rest':
                    for k, pat in REST:
                        if pat in name:
                            vals[k][r['Counter_Name']].append(float(r['Counter_Value']))
                elif 'match_encode_kernel' in name or 'match_encode_sweep_kernel' in name:
                    vals[HEAD[sub]][r['Counter_Name']].append(float(r['Counter_Value']))
path = 'profiles/pmc_traffic.json'
try:
    out = json.load(open(path))
except (OSError, ValueError):
    out = {}
for k, v in vals.items():
    fetch = sum(v['FETCH_SIZE']) / max(1, len(v['FETCH_SIZE'])) * 1024.0
    write = sum(v['WRITE_SIZE']) / max(1, len(v['WRITE_SIZE'])) * 1024.0
    out[k] = {'fetch_size_bytes_raw': round(fetch), 'fetch_bytes_corrected_x2': round(2 * fetch),
              'write_bytes': round(write), 'hbm_bytes_per_launch': round(2 * fetch + write),
              'launches_averaged': len(v['FETCH_SIZE'])}
json.dump(out, open(path, 'w'), indent=1, sort_keys=True)
print(json.dumps({k: out[k] for k in vals}, indent=1, sort_keys=True))
