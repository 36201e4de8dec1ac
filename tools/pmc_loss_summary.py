"""Per-kernel-variant means of the SQ counters collected by tools/pmc_loss.sh."""
import collections
import csv
import glob
import sys

vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/set*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'match_loss_kernel' not in n:
            continue
        key = ('f16' if 'DF16_' in n or '_Float16' in n else 'f32') + \
              ('_bwd' if ('Lb1ELb1ELb1E' in n or 'true, true, true' in n) else '_fwd')
        vals[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(vals):
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(vals[k].items())})
