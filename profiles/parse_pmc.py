#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, TCC slot limits) of
`python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline` into per-kernel HBM bytes per launch.

Correction per /opt/skills/guides/MI355X_MICROARCH.md §HBM: counters are in KiB; on gfx950 FETCH_SIZE reports half of
the bytes of a wide (16 B/lane) coalesced read stream, so fetch bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 is exact
for 16 B/lane streaming stores.  Infinity-Cache hits are included in FETCH_SIZE (it counts L2 fabric requests).

usage: parse_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [source label]
"""
import collections
import csv
import json
import re
import sys


def short(name):
    m = re.search(r'(gemm_nt_mfma_kernel|gemm_tn_mfma_kernel)ILi(\d)ELi(\d)ELi(\d)ELi(\d)E', name)
    if m:
        return f'{m.group(1)}<{m.group(2)},{m.group(3)},{m.group(4)},{m.group(5)}>'
    m = re.search(r'(conv3x3_slab_mfma_kernel)ILi(\d)ELb(\d)E', name)
    if m:
        return f'{m.group(1)}<{m.group(2)},{m.group(3)}>'
    m = re.search(r'(gemm_nt_ring_kernel)ILi(\d)E', name)
    if m:
        return f'{m.group(1)}<{m.group(2)}>'
    m = re.search(r'(conv3x3_wgrad_slab_kernel)ILi(\d)E', name)
    if m:
        return f'{m.group(1)}<{m.group(2)}>'
    m = re.search(r'mlp_chain_pair_kernelI\w+?Li\d+ELi\d+ELi(\d)E', name)
    if m:
        return 'mlp_chain_pair_kernel<fwd>' if m.group(1) == '0' else 'mlp_chain_pair_kernel<bwd>'
    m = re.search(r'mlp_chain_kernelI\w+?Li\d+ELi\d+ELi(\d)ELi\d+E', name)
    if m:
        return 'mlp_chain_kernel<fwd>' if m.group(1) == '0' else 'mlp_chain_kernel<bwd>'
    m = re.search(r'N_1\d+([a-z0-9_]+_kernel)', name)
    if m:
        return m.group(1)
    m = re.search(r'([a-z0-9_]+_kernel)', name)
    return m.group(1) if m else name[:60]


def agg(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        k = short(r['Kernel_Name'])
        d[k][0] += 1
        d[k][1] += float(r['Counter_Value'])
    return d


def main():
    fetch, write = agg(sys.argv[1], 'FETCH_SIZE'), agg(sys.argv[2], 'WRITE_SIZE')
    out = {}
    for k, (n, v) in fetch.items():
        w = write.get(k, [0, 0.0])[1]
        fb, wb = 2.0 * v * 1024.0, w * 1024.0
        out[k] = {'launches': n, 'fetch_bytes': fb, 'write_bytes': wb, 'bytes_per_launch': (fb + wb) / max(n, 1)}
    ranked = sorted(out, key=lambda k: -out[k]['fetch_bytes'] - out[k]['write_bytes'])
    if len(sys.argv) > 4:
        out['_meta'] = {'source': sys.argv[4]}
    json.dump(out, open(sys.argv[3], 'w'), indent=1, sort_keys=True)
    for k in ranked[:12] or sorted(out, key=lambda k: -out[k]['fetch_bytes'] - out[k]['write_bytes'])[:12]:
        o = out[k]
        print(f"{k:40s} launches {o['launches']:4d}  fetch {o['fetch_bytes'] / 1e9:7.2f} GB  write {o['write_bytes'] / 1e9:7.2f} GB"
              f"  per launch {o['bytes_per_launch'] / 1e6:9.1f} MB")


if __name__ == '__main__':
    main()
