"""write profiles/spmv_traffic.json from the two rocprofv3 --pmc passes of
scripts/final_profiles.sh (FETCH_SIZE / WRITE_SIZE, separate passes, SpMV
alone), stamped with the commit and the hashes of the kernel sources the
numbers were taken on (bench.py prints `roofline.traffic` only while they
match)

    python scripts/stamp_traffic.py gpurun_out/<tag>/pmc_summary.json
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

summ = json.load(open(sys.argv[1]))
path = os.path.join(ROOT, 'profiles', 'spmv_traffic.json')
rec = json.load(open(path))


def avg(counter, frag):
    vals = [v for k, v in summ.items()
            if k.startswith(counter + ':') and frag in k]
    return vals[0] if vals else None


def traffic(frag):
    f, w = avg('FETCH_SIZE', frag), avg('WRITE_SIZE', frag)
    if f is None or w is None:
        return None, None, None
    # KB -> bytes; FETCH_SIZE tallies 128-byte requests at 64 on gfx950
    return f, w, (2.0*f + w)*1024.0


f, w, t = traffic('k_spmv_stream16<')
if t:
    rec.update(FETCH_SIZE_kb_raw=f, WRITE_SIZE_kb=w, hbm_bytes_per_launch=t)
f, w, t = traffic('k_spmv_pair16x<')
if t:
    rec.setdefault('pair_format', {}).update(
        FETCH_SIZE_kb_raw=f, WRITE_SIZE_kb=w, hbm_bytes_per_launch=t)
try:
    head = subprocess.check_output(['git', 'rev-parse', 'HEAD'],
                                   cwd=ROOT).decode().strip()
except Exception:
    head = None
rec['sources_sha256'] = bench.kernel_sources_sha256()
rec['taken_at_commit'] = head
rec['summary_file'] = os.path.relpath(os.path.abspath(sys.argv[1]), ROOT)
json.dump(rec, open(path, 'w'), indent=1)
print(json.dumps({k: rec[k] for k in ('hbm_bytes_per_launch', 'sources_sha256',
                                      'taken_at_commit')}, indent=1))
