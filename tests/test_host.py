"""
CPU-side tests (-m "not gpu"): the C ABI loads and exports every symbol of include/degnorm_amd.h, the host
mirror keeps the reference's constructor / error behaviour, helpers keep the reference's semantics, the
result writer produces the reference's files, and the product refuses to run without the HIP device.
"""
import ctypes
import os
import pickle
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_header_symbol():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    from degnorm_amd import _lib, build
    build.build_library()                      # hipcc cross-compiles gfx950 without a GPU
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = ge.header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), 'missing C ABI symbol ' + s
    lib.dn_version.restype = ctypes.c_char_p
    assert b'degnorm_amd' in lib.dn_version()
    # no compute without a GPU: only queries
    for p in range(2, 13):
        assert lib.dn_p_supported(p) == 1
    assert lib.dn_p_supported(50) == 1 and lib.dn_p_supported(64) == 1       # run-time-p kernels
    assert lib.dn_p_supported(1) == 0 and lib.dn_p_supported(65) == 0


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: on a box without a HIP device, opening a Device raises (and names the reason)."""
    from degnorm_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip('a GPU is visible here')
    with pytest.raises(_lib.DegnormAmdError, match='no HIP device'):
        _lib.Device(0)
    from collections import OrderedDict
    from degnorm_amd.nmf import GeneNMFOA
    cov = OrderedDict(a=np.ones((3, 300)), b=np.ones((3, 260)))
    with pytest.raises(_lib.DegnormAmdError):
        GeneNMFOA(degnorm_iter=1).run(cov, np.ones((2, 3)))


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under degnorm_amd/ may reference it."""
    pkg = os.path.join(ROOT, 'degnorm_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in text and 'from oracle' not in text and 'liboracle' not in text, f


def test_constructor_normalisation_matches_reference():
    """GeneNMFOA.__init__ rules (nmf.py:30-53): abs/int, min_high_coverage >= 2 and forced to 2 when downsampling."""
    from degnorm_amd.nmf import GeneNMFOA
    m = GeneNMFOA(degnorm_iter=-3.7, nmf_iter='40', bins=-10, min_high_coverage=1, n_jobs=-2)
    assert (m.degnorm_iter, m.nmf_iter, m.bins, m.min_high_coverage, m.n_jobs) == (3, 40, 10, 2, 2)
    assert m.min_bins == 2.0 and m.downsample_rate == 1 and m.fitted is False
    assert GeneNMFOA(downsample_rate=20, min_high_coverage=77).min_high_coverage == 2
    assert GeneNMFOA(min_high_coverage=77).min_high_coverage == 77
    assert GeneNMFOA.fit is GeneNMFOA.run          # BASELINE.json's `fit` alias
    d = GeneNMFOA()
    assert (d.degnorm_iter, d.nmf_iter, d.bins, d.min_high_coverage, d.random_state) == (5, 100, 20, 50, 123)


def test_split_into_chunks_semantics():
    """utils.py:176-192: chunk size ceil(len/n); may return fewer than n chunks."""
    from degnorm_amd.utils import split_into_chunks, chunk_bounds
    assert [len(c) for c in split_into_chunks(list(range(201)), 20)] == [11] * 18 + [3]
    assert [len(c) for c in split_into_chunks(list(range(19)), 20)] == [1] * 19
    assert [len(c) for c in split_into_chunks(list(range(20000)), 8)] == [2500] * 8
    assert split_into_chunks(list(range(7)), 3) == [[0, 1, 2], [3, 4, 5], [6]]
    assert chunk_bounds(7, 3) == [0, 3, 6, 7]


def test_synth_generator_is_deterministic_and_shardable():
    from degnorm_amd import synth
    a, ca = synth.synth_gene(2, 17, 10)
    b, cb = synth.synth_gene(2, 17, 10)
    np.testing.assert_array_equal(a, b)
    assert ca == cb and a.shape[0] == 10 and 200 <= a.shape[1] <= 5000 and a.min() >= 0
    assert a.shape[1] == synth.gene_length(2, 17)
    packed, lengths, reads, cls = synth.synth_packed(2, [5, 17, 3], 10, n_threads=1)
    assert lengths[1] == a.shape[1]
    o = int(lengths[0]) * 10
    np.testing.assert_array_equal(packed[o:o + a.size].reshape(a.shape), a.astype(np.float32))
    np.testing.assert_array_equal(reads[1], np.round(a.sum(axis=1) / 100.))
    # integer counts < 2^24: the float32 device copy is exact
    assert a.max() < 2 ** 24 and np.array_equal(a, a.astype(np.float32).astype(np.float64))
    # all classes appear in a modest draw
    _, _, classes = synth.synth_dataset(2, 400, 4, 200, 400)
    assert set(classes) == set(range(9))


def test_save_results_files(tmp_path):
    """save_results writes the reference's files (nmf.py:603-711): 3 CSVs + per-chromosome pickles."""
    import pandas as pd
    from degnorm_amd.nmf import GeneNMFOA
    m = GeneNMFOA(degnorm_iter=2)
    m.genes = ['g2', 'g0', 'g1']
    m.p = 2
    m.rho = np.array([[0.1, 0.2], [0.3, 0.4], [0.5, 0.6]])
    m.x_adj = m.rho * 100
    m.ran_baseline_selection = np.array([[True, False], [False, False], [True, True]])
    est = [np.full((2, 4), k, dtype=float) for k in range(3)]
    manifest = pd.DataFrame({'chr': ['chr1', 'chr2', 'chr1', 'chr9'], 'gene': ['g0', 'g1', 'g2', 'zz']})
    with pytest.raises(ValueError, match='Model not yet fit'):
        m.save_results(est, manifest, output_dir=str(tmp_path))
    m.fitted = True
    with pytest.raises(IOError):
        m.save_results(est, manifest, output_dir=str(tmp_path / 'missing'))
    with pytest.raises(ValueError, match='columns `chr` and `gene`'):
        m.save_results(est, manifest[['gene']], output_dir=str(tmp_path))
    with pytest.raises(ValueError, match='sample IDs'):
        m.save_results(est, manifest, output_dir=str(tmp_path), sample_ids=['a'])
    m.save_results(est, manifest, output_dir=str(tmp_path), sample_ids=['s1', 's2'])
    di = pd.read_csv(tmp_path / 'degradation_index_scores.csv')
    assert list(di.columns) == ['chr', 'gene', 's1', 's2']
    assert list(di.gene) == ['g2', 'g0', 'g1'] and list(di.chr) == ['chr1', 'chr1', 'chr2']
    np.testing.assert_allclose(di[['s1', 's2']].values, m.rho)
    adj = pd.read_csv(tmp_path / 'adjusted_read_counts.csv')
    np.testing.assert_allclose(adj[['s1', 's2']].values, m.x_adj)
    ran = pd.read_csv(tmp_path / 'ran_baseline_selection.csv')
    assert list(ran.columns) == ['chr', 'gene', 'iter_0', 'iter_1']
    assert ran.iter_0.tolist() == [True, False, True]
    with open(tmp_path / 'chr1' / 'estimated_coverage_matrices_chr1.pkl', 'rb') as f:
        d = pickle.load(f)
    assert sorted(d) == ['g0', 'g2'] and d['g2'][0, 0] == 0 and d['g0'][0, 0] == 1
    with open(tmp_path / 'chr2' / 'estimated_coverage_matrices_chr2.pkl', 'rb') as f:
        assert list(pickle.load(f)) == ['g1']


def _tier_variants():
    from degnorm_amd import build
    return [(p, nt) for p in build.PAIR_P_LIST for nt in build.NT_LIST] + [(p, -64) for p in build.PAIR_P_LIST]   # -64: the pair build


@pytest.fixture(scope='module')
def tier_isa(tmp_path_factory):
    """ISA of every shipped register-tier translation unit (all p of build.PAIR_P_LIST x {wide, narrow, pair}), compiled in
    parallel with exactly the flags build.py uses (FLAGS + sched_flags(p) + EXTRA)."""
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    from degnorm_amd import build
    d = tmp_path_factory.mktemp('isa')
    src = os.path.join(ROOT, 'degnorm_amd', 'csrc', 'dn_inst.hip')

    def compile_one(v):
        p, nt = v
        out = str(d / 'k_p{0}_{1}.s'.format(p, 'pair' if nt < 0 else nt))
        cmd = [build._hipcc()] + build.FLAGS + build.sched_flags(p) + build.EXTRA + ['-DDN_P={0}'.format(p), '-DDN_NT={0}'.format(abs(nt))] + \
              (['-DDN_PAIR=1'] if nt < 0 else []) + ['-S', '--cuda-device-only', src, '-o', out]
        subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        return v, out
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        return dict(ex.map(compile_one, _tier_variants()))


@pytest.mark.parametrize('p,nt', _tier_variants())
def test_register_tier_registers_are_private(tier_isa, p, nt):
    """
    The register tier (csrc/dn_kernels.hpp) keeps x + lambda in AGPRs through inline v_accvgpr moves with literal register
    numbers, behind the compiler's back.  That is only sound if (1) the compiler itself never allocates an accumulation
    register inside nmf_call (its budget there is 256 architectural VGPRs), (2) nmf_call saves and restores every tier
    register around its body -- the kernel may park values in AGPRs across the call -- and (3) the kernel descriptor gives a
    lane all 256 accumulation registers on top of its architectural ones, which also pins the kernel to one wave per SIMD.
    Checked on the ISA of EVERY shipped tier build (p = 8..12, wide / narrow / pair).
    """
    import re
    out = tier_isa[(p, nt)]
    pair = nt < 0
    nt = abs(nt)
    fn, inasm = None, False
    compiler_agpr = {}          # function -> instructions outside inline asm that name an AGPR
    tier_regs = {}              # function -> AGPR numbers named inside inline asm
    for line in open(out):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            fn = m.group(1)
        t = line.strip()
        if t.startswith(';;#ASMSTART'):
            inasm = True
            continue
        if t.startswith(';;#ASMEND'):
            inasm = False
            continue
        if not t or t[0] in ';.':
            continue
        code = t.split(';')[0]
        regs = [int(x) for x in re.findall(r'\ba(\d+)\b', code)] + [int(x) for pair in re.findall(r'\ba\[(\d+):(\d+)\]', code) for x in pair]
        if not regs:
            continue
        if inasm:
            tier_regs.setdefault(fn, set()).update(regs)
        else:
            compiler_agpr[fn] = compiler_agpr.get(fn, 0) + 1
    call = sorted(f for f in tier_regs if 'nmf_call' in f)
    # the tier is touched in nmf_call only: the function of the four hot bodies and its twin for the safe repeat (SAFE = true)
    assert len(call) == 2 and all('nmf_call' in f for f in tier_regs)
    def cols(regs_per_col):
        return min(12, 256 // regs_per_col) * regs_per_col
    n_tier = max(cols(2 * p), cols(2 * p + (p + 1) // 2))                       # without / with the packed counts
    text = open(out).read()
    regs_of = lambda bs: set(int(x) for b in bs for x in re.findall(r'\ba(\d+)\b', b))
    for fn_call in call:
        assert tier_regs[fn_call] == set(range(n_tier))                         # a0 .. a(n_tier - 1), all of them
        assert compiler_agpr.get(fn_call, 0) == 0                               # (1)
        body = text[text.index(fn_call + ':'):]
        body = body[:body.index('.Lfunc_end')]
        blocks = [b for b in re.findall(r';;#ASMSTART(.*?);;#ASMEND', body, flags=re.S) if 'v_accvgpr' in b]
        assert all('v_accvgpr_read_b32' in b for b in blocks[:n_tier]) and regs_of(blocks[:n_tier]) == set(range(n_tier))     # (2) the save comes first
        # ... and every return of the function (the epilogue is duplicated for the early exits) sits behind a full restore
        runs, cur = [], []
        for b in blocks:
            if 'v_accvgpr_write_b32' in b:
                cur.append(b)
            else:
                if cur:
                    runs.append(cur)
                cur = []
        if cur:
            runs.append(cur)
        restores = [r for r in runs if len(r) >= n_tier and regs_of(r[-n_tier:]) == set(range(n_tier))]
        assert len(restores) >= len(re.findall(r's_setpc_b64 s\[30:31\]', body)) >= 1     # returns (other s_setpc are long branches)
        assert all('v_accvgpr_write_b32' in b for b in blocks[-n_tier:]) and regs_of(blocks[-n_tier:]) == set(range(n_tier))
    if pair:
        # the two wavefronts of a pair workgroup walk different genes: a workgroup barrier anywhere in the kernel would deadlock
        kb = text[text.index('_ZN2dn8nmf_call'):text.index('.amdhsa_kernel _ZN2dn10k_baseline')]
        assert 's_barrier' not in kb
    kern = text[text.index('.amdhsa_kernel _ZN2dn10k_baseline'):]
    # (3) all 256 AGPRs sit behind the architectural registers (a255 is named once, so the allocation reaches it) and the lane's
    # total exceeds 256 registers: one wave per SIMD.  (p = 8, 9 need fewer than 256 architectural VGPRs: accum_offset 248.)
    nfv = int(re.search(r'\.amdhsa_next_free_vgpr (\d+)\b', kern[:4000]).group(1))
    acc = int(re.search(r'\.amdhsa_accum_offset (\d+)\b', kern[:4000]).group(1))
    assert nfv - acc == 256 and nfv > 256 and acc <= 256


def test_pass_of_the_on_chip_body_has_no_scratch_or_scalar_spill_traffic(tmp_path):
    """
    A static guard for the hot loop: at one wave per SIMD a scratch reload inside the T loop costs about 1 % of
    config 2's sweep (measured twice in round 3), and the allocator puts one there at the slightest provocation.  The pass of the
    on-chip body of nmf_call<10, *> -- register tier and LDS tier, between the iter_begin and pass_end marks of a -DDN_MARKS build --
    must contain no scratch access and no v_writelane (scalar spill), and its register tier the 161 vector instructions per column
    of the raw-unit pass (1 610 for the ten columns).
    """
    import re
    import subprocess
    from degnorm_amd import build
    src = os.path.join(ROOT, 'degnorm_amd', 'csrc', 'dn_inst.hip')
    for nt, extra in ((128, []), (64, ['-DDN_PAIR=1'])):
        out = str(tmp_path / 'marks_{0}.s'.format(nt))
        cmd = [build._hipcc()] + build.FLAGS + build.sched_flags(10) + build.EXTRA + ['-DDN_P=10', '-DDN_NT={0}'.format(nt), '-DDN_MARKS'] + extra + \
              ['-S', '--cuda-device-only', src, '-o', out]
        subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        text = open(out).read()
        fn = text[text.index('_ZN2dn8nmf_callILi10ELi{0}ELb0E'.format(nt)):]
        fn = fn[:fn.index('.Lfunc_end')].split('\n')
        marks = [(i, re.search(r'DN_MARK (\w+)', l).group(1)) for i, l in enumerate(fn) if 'DN_MARK' in l]
        i0 = next(i for i, m in marks if m == 'iter_begin')                    # body 0 = the on-chip body
        names = [m for i, m in marks if i >= i0][:5]
        assert names[:4] == ['iter_begin', 'zeroed', 'reg_tier', 'lds_tier'] and names[4] == 'pass_end', names
        i_reg = next(i for i, m in marks if i > i0 and m == 'reg_tier')
        i_lds = next(i for i, m in marks if i > i_reg and m == 'lds_tier')
        i_end = next(i for i, m in marks if i > i_lds and m == 'pass_end')
        ops = [l.strip().split()[0] for l in fn[i0:i_end] if l.startswith('\t') and l.strip() and l.strip()[0] not in '.;']
        assert not [o for o in ops if o.startswith('scratch_') or o == 'v_writelane_b32'], 'spill traffic in the pass (NT = %d)' % nt
        reg_ops = [l.strip().split()[0] for l in fn[i_reg:i_lds] if l.startswith('\t') and l.strip() and l.strip()[0] not in '.;']
        assert sum(o.startswith('v_') for o in reg_ops) == 1610


def test_generated_dpp_ops_header_is_current(tmp_path):
    """degnorm_amd/csrc/dn_dpp_ops.hpp is generated (tools/gen_dpp_ops.py: one inline-assembly statement per product and sample
    count for the round-4 eigen-solver); the committed file must be what the generator writes."""
    import importlib.util
    import shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    committed = open(os.path.join(root, 'degnorm_amd', 'csrc', 'dn_dpp_ops.hpp')).read()
    fake = tmp_path / 'repo'
    (fake / 'tools').mkdir(parents=True)
    (fake / 'degnorm_amd' / 'csrc').mkdir(parents=True)
    shutil.copy(os.path.join(root, 'tools', 'gen_dpp_ops.py'), fake / 'tools' / 'gen_dpp_ops.py')
    spec = importlib.util.spec_from_file_location('gen_dpp_ops_copy', str(fake / 'tools' / 'gen_dpp_ops.py'))
    spec.loader.exec_module(importlib.util.module_from_spec(spec))
    assert open(fake / 'degnorm_amd' / 'csrc' / 'dn_dpp_ops.hpp').read() == committed
    for p in (2, 10, 16):
        assert 'dpp_matvec<%d>' % p in committed and 'dpp_rowdot<%d>' % p in committed
    assert committed.count('row_newbcast:15') == 2                      # only p = 16 reaches the last lane of a row


def test_integration_md_c_loop_compiles_against_the_header(tmp_path):
    """The C-only sharded loop INTEGRATION.md shows (dn_comm_* / dn_init_allreduce / dn_outer_allreduce) is checked against
    include/degnorm_amd.h by the C compiler: names, argument counts and pointer types (-fsyntax-only, no GPU, nothing is run)."""
    import re
    import subprocess
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = [b for b in re.findall(r'```c\n(.*?)```', text, flags=re.S) if 'dn_comm_create' in b]
    assert len(blocks) == 1
    src = tmp_path / 'loop.c'
    src.write_text('#include <stdint.h>\n#include <stddef.h>\n#include "degnorm_amd.h"\n#define P_MAX 64\n'
                   'void bcast(void *buf, size_t bytes, int root);\n'
                   'void sharded_loop(int local_gpu, int r, int R, int64_t n_r, int32_t p, const float *my_packed_f32, const int64_t *my_lengths,\n'
                   '                  const double *my_reads, double *norm, double *scale, double *avg_di, int32_t degnorm_iter, dn_params prm,\n'
                   '                  int32_t *trace, double *rho, double *x_adj, double *x_weighted, uint8_t *ran)\n{\n' + blocks[0] + '}\n')
    r = subprocess.run(['gcc', '-std=c99', '-fsyntax-only', '-Wall', '-Werror=implicit-function-declaration', '-Werror=incompatible-pointer-types',
                        '-Werror=int-conversion', '-I', os.path.join(ROOT, 'include'), str(src)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       universal_newlines=True)
    assert r.returncode == 0, r.stdout


def test_integration_md_ctypes_stub_parses_and_names_exist():
    """The abridged ctypes binding in INTEGRATION.md is valid Python and every dn_* entry point it calls is declared in the header."""
    import ast
    import re
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = [b for b in re.findall(r'```python\n(.*?)```', text, flags=re.S) if 'ctypes.CDLL' in b]
    assert len(blocks) == 1
    ast.parse(blocks[0])
    header = open(os.path.join(ROOT, 'include', 'degnorm_amd.h')).read()
    for name in set(re.findall(r'lib\.(dn_\w+)', blocks[0])):
        assert re.search(r'\b' + name + r'\s*\(', header), name
