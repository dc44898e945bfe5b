"""per-kernel average of the counters in rocprofv3 --pmc output
(`*_counter_collection.csv`), e.g. FETCH_SIZE / WRITE_SIZE in KB

    python scripts/pmc_summary.py 'gpurun_out/pmc_*/*/*counter_collection.csv'
"""
import collections
import csv
import glob
import json
import sys

acc = collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1])):
    for r in csv.DictReader(open(f)):
        acc['{0}:{1}'.format(r['Counter_Name'], r['Kernel_Name'][:64])].append(
            float(r['Counter_Value']))
print(json.dumps({k: sum(v)/len(v) for k, v in acc.items()}, indent=1))
