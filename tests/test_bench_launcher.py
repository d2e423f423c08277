"""
bench.py as the driver starts it for N > 1 (`python3 bench.py --gpus N ...`, no torchrun): the parent must spawn the ranks as
child processes BEFORE touching torch or the GPU, relay rank 0's single JSON line and pass a failure on as a non-zero exit code.
CPU test with a stub in place of `python -m torch.distributed.run` (DN_BENCH_LAUNCHER).
"""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _stub(tmp_path, body):
    path = tmp_path / 'stub_launcher.py'
    path.write_text(textwrap.dedent(body))
    return [sys.executable, str(path)]


def _run_parent(launcher, *argv):
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env['DN_BENCH_LAUNCHER'] = json.dumps(launcher)
    return subprocess.run([sys.executable, BENCH] + list(argv), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          universal_newlines=True, env=env, timeout=120)


def test_parent_relays_rank0_line_and_never_imports_torch(tmp_path):
    launcher = _stub(tmp_path, '''
        import json, os, sys
        # what torchrun would do: run the script once per rank; rank 0 prints the line, everybody prints noise
        script, args = sys.argv[1], sys.argv[2:]
        assert os.path.basename(script) == 'bench.py' and '--gpus' in args and args[args.index('--gpus') + 1] == '4'
        assert os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY') == '0' and os.environ.get('DN_BENCH_PARENT')
        print('rank 1 says hello')
        print('{"not": "the line"}')
        print(json.dumps({'metric': 'genes/sec (20k genes x 10 samples, 5 iters)', 'value': 1.0, 'n_gpus': 4, 'rccl_ranks': 4, 'argv': args}))
        print('trailing noise')
    ''')
    r = _run_parent(launcher, '--gpus', '4', '--steps', '2', '--warmup', '1')
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1                                       # ONE JSON line on stdout, the rest went to stderr
    d = json.loads(lines[0])
    assert d['n_gpus'] == d['rccl_ranks'] == 4 and d['argv'] == ['--gpus', '4', '--steps', '2', '--warmup', '1']
    assert 'rank 1 says hello' in r.stderr and 'trailing noise' in r.stderr


def test_parent_exit_code_follows_the_ranks(tmp_path):
    failing = _stub(tmp_path, '''
        import sys
        print('{"metric": "x", "value": 1}')
        sys.exit(7)
    ''')
    r = _run_parent(failing, '--gpus', '2')
    assert r.returncode == 7 and r.stdout.strip() == ''          # a failed run prints no result line
    silent = _stub(tmp_path, 'print("no result here")\n')
    r = _run_parent(silent, '--gpus', '2')
    assert r.returncode != 0 and r.stdout.strip() == ''


def test_launcher_path_is_torch_free():
    """Importing bench and taking the launcher branch must not import torch (the parent may not initialise the GPU)."""
    code = ('import sys; sys.argv = ["bench.py", "--gpus", "2"]; import bench; '
            'rc = bench.launch_ranks(2, ["--gpus", "2"], launcher=[sys.executable, "-c", '
            '"import json; print(json.dumps(dict(metric=1, n_gpus=2)))"]); '
            'assert rc == 0 and "torch" not in sys.modules, sorted(m for m in sys.modules if m.startswith("torch"))')
    r = subprocess.run([sys.executable, '-c', code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout.strip())['n_gpus'] == 2
