"""Turn the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.sh) into
profiles/pmc_traffic.json: HBM bytes per launch of the dominant kernels.

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM): both counters are in KiB; on gfx950
FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane loads
and LDS-DMA alike) -> doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc_traffic'
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob(os.path.join(root, counter, '*', '*counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name']
            for k, pat in (('match_loss_kernel_fwd', 'match_loss_kernel<float, false'),
                           ('match_loss_kernel_fwd_bwd', 'match_loss_kernel<float, true'),
                           ('match_loss_kernel_f16_fwd', 'match_loss_kernelIDF16_Lb0'),     # rocprofv3 leaves _Float16
                           ('match_loss_kernel_f16_bwd', 'match_loss_kernelIDF16_Lb1'),     # instantiations mangled
                           ('match_encode_kernel', 'match_encode_kernel'), ('score_scan_kernel', 'score_scan_kernel'),
                           ('nms_kernel', 'nms_kernel'), ('finalize_kernel', 'finalize_kernel'),
                           ('cand_partition_kernel', 'cand_partition_kernel')):
                if pat in name:
                    vals[k][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, v in vals.items():
    fetch = sum(v['FETCH_SIZE']) / max(1, len(v['FETCH_SIZE'])) * 1024.0
    write = sum(v['WRITE_SIZE']) / max(1, len(v['WRITE_SIZE'])) * 1024.0
    out[k] = {'fetch_size_bytes_raw': round(fetch), 'fetch_bytes_corrected_x2': round(2 * fetch),
              'write_bytes': round(write), 'hbm_bytes_per_launch': round(2 * fetch + write),
              'launches_averaged': len(v['FETCH_SIZE'])}
json.dump(out, open('profiles/pmc_traffic.json', 'w'), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
