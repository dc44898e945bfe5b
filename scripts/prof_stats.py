"""print the top rows of a rocprofv3 kernel_stats.csv (helper for profiles/)"""
import csv
import glob
import sys

pat = sys.argv[1] if len(sys.argv) > 1 else '*/*kernel_stats.csv'
top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
f = sorted(glob.glob(pat, recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
print(f)
for r in rows[:top]:
    print('{0:58s} calls {1:>6s} tot_us {2:9.1f} avg {3:7.2f} min {4:6.2f} '
          'max {5:8.2f} pct {6}'.format(
              r['Name'][:58], r['Calls'], float(r['TotalDurationNs'])/1e3,
              float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3,
              float(r['MaxNs'])/1e3, r['Percentage']))
