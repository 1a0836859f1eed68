"""`python bench.py --gpus N` starts the N ranks itself: rehearsed on
the CPU with --backend gloo --plan-only (rank processes, rendezvous on 127.0.0.1, channel
blocks per rank, reductions; no kernels)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env_extra=None):
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *argv],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_self_launch_two_ranks_c5_shape():
    line = _run('--gpus', '2', '--backend', 'gloo', '--plan-only', '--workload', 'c5',
                '--channels', '3', '--points', '400000')      # 4000 samples per pulse: the lean tier
    assert line['n_gpus'] == 2 and line['plan_only'] is True
    assert [r['channels'] for r in line['ranks']] == [[0, 3], [3, 6]]     # global channel index
    assert all(r['fused_terms'] > 0 for r in line['ranks'])
    assert line['kernel'].startswith('wfk_sample_lean<double,false')


def test_external_launcher_env_is_honoured():
    # the driver's torchrun form: rendezvous in the environment, --gpus only informative
    line = _run('--gpus', '1', '--plan-only', '--workload', 'c3', '--channels', '2',
                '--points', '20000')
    assert line['n_gpus'] == 1 and line['ranks'][0]['channels'] == [0, 2]


def test_c5_default_shape_is_512_channels_per_rank():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.default_shape('c5') == (512, 10**7)
    assert bench.default_shape('sampler256') == (256, 10**7)
    assert bench.default_shape('c3') == (256, 10**6)


def _launch_raw(*argv, env_extra=None):
    import time
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(env_extra or {})
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *argv],
                       capture_output=True, text=True, env=env, timeout=120)
    return r, time.time() - t0


def test_a_dying_rank_ends_the_job_quickly():
    # rank 1 exits with code 7 before the rendezvous: rank 0 would sit in init_process_group until the
    # collective timeout (minutes); the parent must stop it and return non-zero within ~30 s
    r, dt = _launch_raw('--gpus', '2', '--backend', 'gloo', '--plan-only', '--workload', 'c3',
                        '--channels', '2', '--points', '20000', env_extra={'WFK_BENCH_FAIL_RANK': '1'})
    assert r.returncode != 0 and dt < 30, (r.returncode, dt)
    assert 'rank 1 exited with code 7' in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]      # no result line from a failed job


def test_a_hanging_rank_hits_the_deadline():
    r, dt = _launch_raw('--gpus', '2', '--backend', 'gloo', '--plan-only', '--workload', 'c3',
                        '--channels', '2', '--points', '20000', '--deadline', '8',
                        env_extra={'WFK_BENCH_HANG_RANK': '1'})
    assert r.returncode != 0 and dt < 30, (r.returncode, dt)
    assert 'deadline' in r.stderr


def test_awg_workload_shape_and_tiling():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.default_shape('awg') == (2048, 10**5) and bench.TILE['awg'] == 128
    line = _run('--plan-only', '--workload', 'awg', '--channels', '256')
    assert line['kernel'].startswith('wfk_sample_short<double') and line['ranks'][0]['channels'] == [0, 256]
