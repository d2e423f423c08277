"""
In-tree build of libdegnorm_amd.so (HIP kernels + C ABI) for gfx950 with hipcc.

    python -m degnorm_amd.build [--force]

One translation unit per sample count p (dn_inst.hip -DDN_P=p) plus the C-ABI unit; objects are compiled
in parallel and linked into degnorm_amd/libdegnorm_amd.so.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
# DN_BUILD_TAG=<name> (experiments): objects in csrc/obj_<name>, library in build_variants/lib_<name>.so; load it with
# DN_LIB_PATH=build_variants/lib_<name>.so.  Without a tag: the product library degnorm_amd/libdegnorm_amd.so.
TAG = os.environ.get('DN_BUILD_TAG', '')
OBJ = os.path.join(CSRC, 'obj_' + TAG if TAG else 'obj')
LIB = os.path.join(HERE, '..', 'build_variants', 'lib_{0}.so'.format(TAG)) if TAG else os.path.join(HERE, 'libdegnorm_amd.so')
P_LIST = list(range(2, int(os.environ.get('DN_P_MAX_TEMPLATED', 64)) + 1))   # keep in sync with DN_FOR_EACH_P in csrc/dn_api.hip
ARCH = 'gfx950'
WIDE_NT = int(os.environ.get('DN_WIDE_NT', 256))     # wide-class workgroup size (csrc/dn_api.hip DN_WIDE_NT)
NT_LIST = (WIDE_NT, 128)
# pair build (one wavefront per gene, two genes per 128-thread workgroup): where the register tier exists (dn_kernels.hpp
# DN_RT_MIN_P .. DN_RT_MAX_P; dn_inst.hip refuses to compile a pair unit without it).  The list is handed to dn_api.hip as the
# DN_P_PAIR X-macro, so the dispatcher there cannot drift from what is compiled here.
PAIR_P_LIST = [q for q in P_LIST if int(os.environ.get('DN_RT_MIN_P', 2)) <= q <= int(os.environ.get('DN_RT_MAX_P', 12))]
EXTRA = ['-D' + d for d in os.environ.get('DN_DEFINES', '').split() if d]   # e.g. DN_DEFINES='DN_STAMP=1'
for _k in ('DN_RT_MIN_P', 'DN_RT_MAX_P'):             # the same bounds reach the kernels that PAIR_P_LIST was derived from
    if _k in os.environ:
        EXTRA.append('-D{0}={1}'.format(_k, int(os.environ[_k])))
# kernel translation units: machine-scheduler strategy (DN_HIPCC_FLAGS).  Round 2 shipped -mllvm -amdgpu-sched-strategy=max-ilp
# (+0.7 % on config 2 then); with the raw-unit pass of round 3 the default scheduler is the faster one (283.7 against 285.8-286.3 ms
# per sweep on one box; iterative-minreg 302.6, iterative-ilp crashes the compiler on these units)
SCHED = os.environ.get('DN_HIPCC_FLAGS', '').split()
MAX_ILP = ['-mllvm', '-amdgpu-sched-strategy=max-ilp']


def sched_flags(p):
    """Scheduler flags of the translation units of sample count p: the default scheduler for the register-tier cohorts with the
    raw-unit pass (p = 3 .. 12: +2 ... +4.6 % genes/s per outer iteration, tools/p_sweep.py), max-ILP elsewhere (p = 2: +1.5 %,
    p = 13: +0.7 %, p = 16: +2.5 % with it).  DN_HIPCC_FLAGS overrides both."""
    if 'DN_HIPCC_FLAGS' in os.environ:
        return SCHED
    return [] if 3 <= p <= 12 else MAX_ILP
FLAGS = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-fno-fast-math', '-ffp-contract=on',
         '-Wall', '-Wno-unused-function']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def build_stamp(hipcc):
    """Compiler version and the flags the kernels are compiled with: what dn_version() reports (two builds of one source tree with
    another compiler or other code-generation flags are different binaries; results are reproducible per binary)."""
    try:
        ver = [l for l in _run([hipcc, '--version']).splitlines() if 'clang version' in l or 'HIP version' in l]
    except Exception:
        ver = ['unknown compiler']
    kern = ' '.join(SCHED) if 'DN_HIPCC_FLAGS' in os.environ else 'default scheduler for p = 3..12, max-ilp elsewhere'
    return '; '.join(v.strip() for v in ver) + '; flags: ' + ' '.join(FLAGS[1:]) + '; kernel units: ' + kern + (('; defines: ' + ' '.join(EXTRA)) if EXTRA else '')


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    if r.returncode != 0:
        raise RuntimeError('build step failed: {0}\n{1}'.format(' '.join(cmd), r.stdout))
    return r.stdout


def build_library(force=False, verbose=False):
    """Compile and link; returns the path of the shared library."""
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(os.path.abspath(LIB)), exist_ok=True)
    hipcc = _hipcc()
    hdr = [os.path.join(CSRC, 'dn_kernels.hpp'), os.path.join(CSRC, 'dn_reduce.hpp'), os.path.join(HERE, '..', 'include', 'degnorm_amd.h'),
           os.path.abspath(__file__)]
    jobs = []
    objs = []
    inst = os.path.join(CSRC, 'dn_inst.hip')
    for p in P_LIST:
        for nt in NT_LIST:      # wide-gene class (one 256-thread workgroup per CU) and narrow-gene class (two of 128)
            o = os.path.join(OBJ, 'dn_inst_p{0}_nt{1}.o'.format(p, nt))
            objs.append(o)
            if force or _newer(o, [inst] + hdr):
                jobs.append([hipcc] + FLAGS + sched_flags(p) + EXTRA + ['-DDN_P={0}'.format(p), '-DDN_NT={0}'.format(nt), '-c', inst, '-o', o])
    for p in PAIR_P_LIST:
        o = os.path.join(OBJ, 'dn_inst_p{0}_pair.o'.format(p))
        objs.append(o)
        if force or _newer(o, [inst] + hdr):
            jobs.append([hipcc] + FLAGS + sched_flags(p) + EXTRA + ['-DDN_P={0}'.format(p), '-DDN_NT=64', '-DDN_PAIR=1', '-c', inst, '-o', o])
    gen = os.path.join(CSRC, 'dn_generic.hip')
    for gnt in (256, 64):       # general run-time-p family / one-wavefront-per-gene family (down-sampled regime)
        o_gen = os.path.join(OBJ, 'dn_generic_nt{0}.o'.format(gnt))
        objs.append(o_gen)
        if force or _newer(o_gen, [gen] + hdr):
            jobs.append([hipcc] + FLAGS + EXTRA + ['-DDN_GEN_NT={0}'.format(gnt), '-c', gen, '-o', o_gen])
    asm = os.path.join(CSRC, 'dn_assemble.hip')
    o_asm = os.path.join(OBJ, 'dn_assemble.o')
    objs.append(o_asm)
    if force or _newer(o_asm, [asm] + hdr):
        jobs.append([hipcc] + FLAGS + ['-c', asm, '-o', o_asm])
    api = os.path.join(CSRC, 'dn_api.hip')
    o_api = os.path.join(OBJ, 'dn_api.o')
    objs.append(o_api)
    stamp = build_stamp(hipcc)
    stamp_file = os.path.join(OBJ, 'build_stamp.txt')
    if not os.path.exists(stamp_file) or open(stamp_file).read() != stamp:      # another compiler or other flags: the C-ABI unit names them
        with open(stamp_file, 'w') as f:
            f.write(stamp)
    if force or _newer(o_api, [api, stamp_file] + hdr):
        jobs.append([hipcc] + FLAGS + ['-DDN_BUILD_STAMP="{0}"'.format(stamp.replace('"', "'")),
                                       '-DDN_WIDE_NT={0}'.format(WIDE_NT), '-DDN_P_MAX_TEMPLATED={0}'.format(P_LIST[-1]),
                                       '-DDN_P_PAIR(X)=' + ' '.join('X({0})'.format(q) for q in PAIR_P_LIST), '-c', api, '-o', o_api])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for out in ex.map(_run, jobs):
                if verbose and out.strip():
                    print(out)
    if force or jobs or _newer(LIB, objs):
        _run([hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs + ['-ldl'])
    return LIB


if __name__ == '__main__':
    path = build_library(force='--force' in sys.argv, verbose=True)
    print(path)
