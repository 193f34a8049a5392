"""Data parallelism of the fused step on the GPU: two fresh child processes (one rank each, gloo, both on cuda:0; the
children are created before they touch the GPU) run FusedTrainStep + StepExchange on their own batches. The reduced
flat gradient must equal the single-process gradient of the union of both ranks' batches, and the replicas must agree
bit for bit; with row-sparse tables, also after an optimiser step."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _gpus():
    import torch
    return torch.cuda.device_count()          # (counting devices does not initialise the GPU: children may still be started)


CASES = [(0, 'rows', 'mp', 'pack'), (1, 'rows', 'mp', 'pack'), (0, 'rows', 'mp', 'step'), (1, 'rows', 'mp', 'step'),
         (0, 'dense', 'mp', 'step'), (0, 'dense', 'targetmlp', 'step'), (0, 'dense', 'concat', 'step'),
         # the MLP readout on the chain form: its Linear layers ride in the bucket, the row-sparse tables in the row exchange
         (0, 'dense', 'mlp', 'step'), (1, 'rows', 'mlp', 'step')]


@pytest.mark.parametrize('sparse,tables,readout,touch', CASES)
def test_two_ranks_equal_single_process(tmp_path, sparse, tables, readout, touch):
    _two_ranks(tmp_path, sparse, tables, readout, touch, 'gloo')


@pytest.mark.skipif(_gpus() < 2, reason='RCCL needs one GPU per rank: fewer than 2 GPUs visible on this box')
@pytest.mark.parametrize('sparse,tables,readout,touch', [CASES[1], CASES[3], CASES[4]])
def test_two_ranks_equal_single_process_rccl(tmp_path, sparse, tables, readout, touch):
    """The same over RCCL (backend 'nccl' on ROCm), one rank per GPU: dist.all_reduce / all_gather_into_tensor on device
    buffers (parallel.py: _all_reduce / _all_gather). Collected everywhere, runs the moment two GPUs are visible."""
    _two_ranks(tmp_path, sparse, tables, readout, touch, 'nccl')


@pytest.mark.parametrize('sparse,tables,readout,touch', [CASES[4], CASES[3]])
def test_two_ranks_peer_mapped_exchange(tmp_path, sparse, tables, readout, touch):
    """The bucket summed by the library's one-hop exchange over IPC-mapped peer buffers (csrc/p2p.hip) instead of the
    all-reduce: two processes on the ONE GPU of the box map each other's communication buffers (hipIpcGetMemHandle /
    hipIpcOpenMemHandle work between processes of one device); same assertions as above -- the reduced gradient is the
    single-process gradient of both ranks' batches, the replicas agree bit for bit."""
    _two_ranks(tmp_path, sparse, tables, readout, touch, 'gloo', transport='p2p')


@pytest.mark.parametrize('sparse,tables,readout,touch', [CASES[4], CASES[3]])
def test_two_ranks_one_failed_in_step_sort_is_recovered(tmp_path, sparse, tables, readout, touch):
    """Rank 1's in-step touch-plan sort is forced to give up (mpqe_debug_option TSORT_FAIL): its step leaves the entity-table
    rows unsummed. The data-parallel loop runs its steps checked -- run(checked=True) rebuilds the plan with the library sort
    and sums the rows again before the exchange --, so the reduced gradient is still the single-process one and the replicas
    agree bit for bit; with the row exchange, the recovered rank keeps to the cached in-step exchange plan like its peer."""
    _two_ranks(tmp_path, sparse, tables, readout, touch, 'gloo', fail_sort_rank=1)


def _two_ranks(tmp_path, sparse, tables, readout, touch, backend, transport='rccl', fail_sort_rank=-1):
    world, port = 2, _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY='0', MPQE_DP_BACKEND=backend, MPQE_DP_FAIL_SORT_RANK=str(fail_sort_rank))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'dp_worker.py'), str(tmp_path), str(sparse), tables,
                                       readout, touch, transport], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    r = [np.load(os.path.join(str(tmp_path), 'rank%d.npz' % k)) for k in range(world)]
    ref = r[0]['ref']
    nt = int(r[0]['table_floats'][0])               # the tables are the first parameters of the flat buffer
    # layer / mode parameters: dense, everywhere
    np.testing.assert_allclose(r[0]['flat'][nt:], ref[nt:], rtol=1e-4, atol=2e-6)
    np.testing.assert_array_equal(r[0]['flat'][nt:], r[1]['flat'][nt:])
    # entity tables: the rows some rank touched hold the sum; the others are zero (dense mode) or untouched (sparse)
    D = 64
    tab0, tab1, tref = r[0]['flat'][:nt].reshape(-1, D), r[1]['flat'][:nt].reshape(-1, D), ref[:nt].reshape(-1, D)
    touched = np.abs(tref).sum(axis=1) > 0
    np.testing.assert_allclose(tab0[touched], tref[touched], rtol=1e-4, atol=2e-6)
    np.testing.assert_array_equal(tab0[touched], tab1[touched])
    if sparse:
        assert (tab0[~touched] == 3.0).all() and (tab1[~touched] == 3.0).all()      # nobody wrote them
        np.testing.assert_array_equal(r[0]['params'], r[1]['params'])               # replicas after the optimiser step
    else:
        assert (tab0[~touched] == 0).all() and (tab1[~touched] == 0).all()
    if tables == 'rows':
        np.testing.assert_array_equal(r[0]['union_keys'], r[1]['union_keys'])
    if not (tables == 'rows' and touch == 'step'):      # (in-step plans send fixed-size key / row slots: one per looked-up id --
        # more than this 60-entity KG's whole tables; the sizes it is for have 10^5 .. 10^6 rows)
        assert r[0]['wire'][0] <= r[0]['dense'][0], (r[0]['wire'], r[0]['dense'])   # (tiny KG, 8 relations: most are in the union)
    assert str(r[0]['form'][0]) == str(r[1]['form'][0])
    if transport == 'p2p':
        assert str(r[0]['form'][0]) == 'p2p'


@pytest.mark.skipif(_gpus() < 2, reason='RCCL needs one GPU per rank: fewer than 2 GPUs visible on this box')
def test_bench_two_gpus_rccl():
    """`python bench.py --gpus 2` over RCCL, one rank per GPU, exactly as the driver's scaling run starts it."""
    test_bench_gpus_flag_starts_its_ranks(backend='nccl')


def test_bench_gpus_flag_starts_its_ranks(backend='gloo'):
    """`python bench.py --gpus 2` with no launcher around it must start two ranks itself (a child torch.distributed.run,
    before the parent touches the GPU) and report n_gpus = 2 with the exchange record; here over gloo, both ranks on the
    one GPU of the box (RCCL refuses two ranks on one device)."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--backend', backend, '--steps', '2',
                        '--warmup', '1', '--repeats', '1', '--no-cpu-baseline', '--no-scatter'], env=env, cwd=root,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak'
    assert d['config']['global_query_graphs_per_step'] == 2 * 11 * 512
    assert 'exchange' in d and d['exchange']['bytes_per_rank_per_step'] > 0
    assert 'exchange_note' not in d, d.get('exchange_note')        # the exchange equals the dense all-reduce
    # more ranks than GPUs with the RCCL backend: a refusal, not a one-rank run that claims more
    q = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '64', '--steps', '1', '--warmup', '0'],
                       env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert q.returncode != 0 and b'n_gpus' not in q.stdout
