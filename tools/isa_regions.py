"""
Static instruction budget of one inner NMF-OA iteration, from the ISA of a diagnostic build with region marks (-DDN_MARKS:
`; DN_MARK <name>` comment lines, csrc/dn_kernels.hpp).  For every nmf() body (the four instantiations of nmf_body inside
nmf_call) it lists, per region of the T loop, the instructions by kind -- vector ALU (fp64 arithmetic, AGPR moves, conversions,
cross-lane moves, other), matrix, LDS, global / scratch, scalar -- and prices the region at the measured single-wave issue cadence
(tools/ubench/instr_cost.hip: 4.08 cycles per vector instruction of any kind, 64 cycles per v_mfma_f64_16x16x4).
usage: python tools/isa_regions.py <p> <nt> [pair]      (compiles csrc/dn_inst.hip itself; no GPU needed)
"""
import os, re, subprocess, sys, tempfile
from collections import Counter, OrderedDict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from degnorm_amd import build

CADENCE, MFMA_CYCLES = 4.08, 64.0          # tools/ubench/instr_cost.hip: every vector instruction, one wave per SIMD


def kind(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_accvgpr'): return 'agpr_move'
    if op.startswith(('v_fma_f64', 'v_fmac_f64', 'v_mul_f64', 'v_add_f64', 'v_max_f64', 'v_min_f64', 'v_rsq_f64', 'v_rcp_f64', 'v_sqrt_f64', 'v_cmp')) and 'f64' in op: return 'fp64'
    if op.startswith('v_cvt'): return 'convert'
    if 'permlane' in op or 'dpp' in op or op.startswith(('v_readlane', 'v_readfirstlane', 'v_writelane')): return 'cross_lane'
    if op.startswith('v_'): return 'valu_other'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'scratch_', 'flat_', 'buffer_')): return 'memory'
    return 'scalar'


def main():
    p, nt = int(sys.argv[1]), int(sys.argv[2])
    pair = len(sys.argv) > 3 and sys.argv[3] == 'pair'
    out = os.path.join(tempfile.mkdtemp(), 'k.s')
    cmd = [build._hipcc()] + build.FLAGS + build.sched_flags(p) + build.EXTRA + ['-DDN_P=%d' % p, '-DDN_NT=%d' % nt, '-DDN_MARKS'] + (['-DDN_PAIR=1'] if pair else []) + \
          ['-S', '--cuda-device-only', os.path.join(ROOT, 'degnorm_amd', 'csrc', 'dn_inst.hip'), '-o', out]
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    text = open(out).read()
    fn = text[text.index('_ZN2dn8nmf_call'):]
    fn = fn[:fn.index('.Lfunc_end')]
    lines = fn.split('\n')
    # split the function into the bodies: each T loop starts at an "iter_begin" mark
    starts = [i for i, l in enumerate(lines) if 'DN_MARK iter_begin' in l]
    print('nmf_call<%d,%d>%s: %d nmf() bodies with a T loop' % (p, nt, ' (pair build)' if pair else '', len(starts)))
    for b, lo in enumerate(starts):
        hi = starts[b + 1] if b + 1 < len(starts) else len(lines)
        regions, cur = OrderedDict(), 'iter_begin'
        for l in lines[lo:hi]:
            m = re.search(r'DN_MARK (\w+)', l)
            if m:
                cur = m.group(1)
                if cur == 'solved':
                    break
                continue
            t = l.strip()
            if not l.startswith('\t') or not t or t[0] in '.;':
                continue
            regions.setdefault(cur, Counter())[kind(t.split()[0])] += 1
        print('-- body %d' % b)
        tot_v, tot_m = 0, 0
        for name, c in regions.items():
            v = c['fp64'] + c['agpr_move'] + c['convert'] + c['cross_lane'] + c['valu_other']
            tot_v += v; tot_m += c['mfma']
            print('   after %-16s vector %5d (fp64 %4d, agpr moves %4d, convert %3d, cross-lane %3d, other %3d)  mfma %2d  lds %3d  memory %3d  scalar %4d   ~%6.0f cycles'
                  % (name, v, c['fp64'], c['agpr_move'], c['convert'], c['cross_lane'], c['valu_other'], c['mfma'], c['lds'], c['memory'], c['scalar'],
                     v * CADENCE + c['mfma'] * MFMA_CYCLES))
        print('   total vector %d, mfma %d -> ~%.0f issue cycles per inner iteration (runtime loops of the LDS / spill tiers and of the solver counted once)'
              % (tot_v, tot_m, tot_v * CADENCE + tot_m * MFMA_CYCLES))


if __name__ == '__main__':
    main()
