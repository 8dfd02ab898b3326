"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs into profiles/<tag>_pmc_traffic.json.

  python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

Per kernel: average over launches of FETCH_SIZE and WRITE_SIZE (reported in KiB), and the HBM bytes per
launch with the gfx950 correction for wide (16 B/lane) reads: FETCH_SIZE counts 64 B per 128-B request,
so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact for 16-B-per-lane stores."""
import collections
import csv
import json
import sys


def avg_by_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def main(fetch_csv, write_csv, out_json):
    f, n = avg_by_kernel(fetch_csv, 'FETCH_SIZE')
    w, _ = avg_by_kernel(write_csv, 'WRITE_SIZE')
    out = {}
    for k in sorted(set(f) | set(w)):
        if not k.startswith('void ctr::') and not k.startswith('ctr::'):
            continue
        fk, wk = f.get(k, 0.0), w.get(k, 0.0)
        out[k] = dict(launches=n.get(k, 0), fetch_size_kib=fk, write_size_kib=wk,
                      hbm_bytes_per_launch=int(2 * fk * 1024 + wk * 1024),
                      note='read bytes = 2 x FETCH_SIZE (gfx950 counts 64 B per 128-B request); write bytes = WRITE_SIZE')
    json.dump(out, open(out_json, 'w'), indent=1)
    print('wrote', out_json, len(out), 'kernels')


if __name__ == '__main__':
    main(*sys.argv[1:4])
