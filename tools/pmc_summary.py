"""
Condenses a tools/profile_round.sh output directory into the files kept under profiles/<round>/:
per-kernel FETCH_SIZE / WRITE_SIZE summaries, the kernel stats, and pmc_traffic.json (HBM-side bytes per launch of the
dominant kernel, FETCH_SIZE doubled as calibrated in profiles/round1/pmc_calibration_*.csv and prescribed for gfx950 by
the MI355X guide).  The json names the kernel sources (bench.source_hash) and the library binary it was measured
on; bench.py refuses to report the traffic for any other source tree.
Usage: python tools/pmc_summary.py gpurun_out/<dir> profiles/round2 <prefix> [config]
"""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict


def per_kernel(path, counter):
    rows = defaultdict(lambda: defaultdict(float))        # kernel -> dispatch -> summed value
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r['Counter_Name'] == counter:
                rows[r['Kernel_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
    return {k: list(v.values()) for k, v in rows.items()}


def write_summary(per, out):
    with open(out, 'w') as fh:
        fh.write('kernel,dispatches,mean_KiB,min_KiB,max_KiB\n')
        for k, v in per.items():
            fh.write('"{0}",{1},{2:.3f},{3:.3f},{4:.3f}\n'.format(k, len(v), sum(v) / len(v), min(v), max(v)))


def spaced(name):
    """k_baseline<10,256> -> k_baseline<10, 256> (how rocprofv3 prints template arguments)."""
    inner = name[name.index('<') + 1:-1].split(',')
    return name[:name.index('<') + 1] + ', '.join(inner) + '>'


def mean_of(per, needle):
    for k, v in per.items():
        if needle in k:
            return sum(v) / len(v)
    return 0.0


def main():
    src, dst, prefix = sys.argv[1], sys.argv[2], sys.argv[3]
    config = sys.argv[4] if len(sys.argv) > 4 else 'c2'
    os.makedirs(dst, exist_ok=True)
    fetch = per_kernel(os.path.join(src, 'fetch', 'run_counter_collection.csv'), 'FETCH_SIZE')
    write = per_kernel(os.path.join(src, 'write', 'run_counter_collection.csv'), 'WRITE_SIZE')
    write_summary(fetch, os.path.join(dst, prefix + 'pmc_fetch_size_per_kernel.csv'))
    write_summary(write, os.path.join(dst, prefix + 'pmc_write_size_per_kernel.csv'))
    shutil.copy(os.path.join(src, 'stats', 'run_kernel_stats.csv'), os.path.join(dst, prefix + 'kernel_stats.csv'))
    with open(os.path.join(src, 'stats', 'run_kernel_trace.csv')) as fh, \
            open(os.path.join(dst, prefix + 'kernel_trace_baseline.csv'), 'w') as out:
        for i, line in enumerate(fh):
            if i == 0 or 'dn::k_' in line:
                out.write(line)
    shutil.copy(os.path.join(src, 'bench.json'), os.path.join(dst, prefix + 'bench.json'))

    bench = json.load(open(os.path.join(src, 'bench.json')))
    hashes = json.load(open(os.path.join(src, 'hashes.json')))
    traffic = {
        'method': 'tools/profile_round.sh: rocprofv3 --kernel-trace --pmc <counter> --output-format csv -- python3 bench.py '
                  '--warmup 0 --cpu-sample 0 --parity-genes 0 (separate passes for FETCH_SIZE and WRITE_SIZE; values in KiB; FETCH_SIZE '
                  'doubled: on gfx950 it reports 1/2 of the bytes of coalesced reads -- MI355X guide, and calibrated in round 1 with '
                  'tools/ubench/stream_read.hip: 0.500 for 4-B and 8-B-per-lane reads, WRITE_SIZE 1.000).  Fabric-side counters: '
                  'Infinity-Cache hits are included.',
        'config': config,
        'source_sha256': hashes['source_sha256'],
        'lib_sha256': hashes['lib_sha256'],
        'split_length': bench['roofline'].get('split_length'),
        'pair_length': bench['roofline'].get('pair_length'),
        'kernels': {},
    }
    names = [(bench['roofline'].get('kernel'), bench['roofline'].get('genes_in_kernel'))]
    for key in ('second_kernel', 'concurrent_kernel', 'iteration_kernel'):
        k = bench['roofline'].get(key)
        if k and k.get('kernel'):
            names.append((k['kernel'], k.get('genes')))
    for k in bench['roofline'].get('concurrent_kernels') or []:
        if k.get('kernel'):
            names.append((k['kernel'], k.get('genes')))
    whole = (bench.get('config') or {}).get('genes_per_gpu')              # kernels that take every gene of the shard (config 4)
    for name, genes in names:
        if not name:
            continue
        genes = whole if genes is None else genes
        rp = spaced(name) if '<' in name else name
        f_k, w_k = mean_of(fetch, rp), mean_of(write, rp)
        traffic['kernels'][name] = {
            'genes_in_kernel': genes,
            'FETCH_SIZE_KiB_per_launch': f_k, 'WRITE_SIZE_KiB_per_launch': w_k,
            'read_bytes_per_launch': 2.0 * f_k * 1024.0, 'write_bytes_per_launch': w_k * 1024.0,
            'hbm_bytes_per_launch': 2.0 * f_k * 1024.0 + w_k * 1024.0,
        }
    with open(os.path.join(dst, 'pmc_traffic_{0}.json'.format(config)), 'w') as fh:
        json.dump(traffic, fh, indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == '__main__':
    main()
