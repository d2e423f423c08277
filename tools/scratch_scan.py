"""Which loops of a kernel translation unit touch scratch (register spills, callee-saved saves)?  Compiles the unit to ISA with the
product's flags (no GPU needed) and lists, per function, every loop that contains scratch_load / scratch_store, with the lines of
those inside the long (T) loops.  Round 4 used it to find the reloads in front of the spill tier's first loads (csrc/dn_kernels.hpp,
spill_tier) and the scratch traffic of the wide-cohort initial pass (csrc/dn_generic.hip).
usage: python tools/scratch_scan.py <p> <nt> [pair]          templated unit dn_inst.hip -DDN_P=<p> -DDN_NT=<nt> [-DDN_PAIR=1]
       python tools/scratch_scan.py generic <nt>             the run-time-p family dn_generic.hip -DDN_GEN_NT=<nt>"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from degnorm_amd import build


def main():
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    csrc = os.path.join(ROOT, 'degnorm_amd', 'csrc')
    if sys.argv[1] == 'generic':
        src, defs, sched = os.path.join(csrc, 'dn_generic.hip'), ['-DDN_GEN_NT=' + sys.argv[2]], []
    else:
        p = int(sys.argv[1])
        src, defs, sched = os.path.join(csrc, 'dn_inst.hip'), ['-DDN_P=%d' % p, '-DDN_NT=' + sys.argv[2]], build.sched_flags(p)
        if 'pair' in sys.argv[3:]:
            defs.append('-DDN_PAIR=1')
    out = os.path.join(tempfile.mkdtemp(), 'unit.s')
    cmd = [build._hipcc()] + [f for f in build.FLAGS if not f.startswith('-W')] + sched + defs + ['-S', '--cuda-device-only', src, '-o', out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split('\n')
    funcs = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
    ends = [i for i, l in enumerate(lines) if l.startswith('.Lfunc_end')]
    for i, name in funcs:
        body = lines[i:min(x for x in ends if x > i)]
        total = sum('scratch_' in l for l in body)
        print('%s: %d lines, %d scratch accesses' % (name[:100], len(body), total))
        for h in [k for k, l in enumerate(body) if 'Loop Header' in l]:
            k = h
            while not body[k].startswith('.LBB'):
                k -= 1
            label, end = body[k].split(':')[0].strip(), None
            for j in range(h + 1, len(body)):
                if re.match(r's_c?branch\w*\s+' + re.escape(label) + r'$', body[j].split(';')[0].strip()):
                    end = j
            if end is None:
                continue
            inside = [(k + q, l.strip()) for q, l in enumerate(body[k:end + 1]) if 'scratch_' in l]
            if inside:
                print('    loop at line %d (%d lines, %s): %d scratch accesses' % (k, end - k, body[h].split('=>')[-1].strip(), len(inside)))
                if end - k > 1000:
                    for ln, txt in inside:
                        print('        %d  %s' % (ln, txt[:100]))


if __name__ == '__main__':
    main()
