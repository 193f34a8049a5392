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


@pytest.mark.parametrize('sparse,tables', [(0, 'rows'), (1, 'rows'), (0, 'dense')])
def test_two_ranks_equal_single_process(tmp_path, sparse, tables):
    world, port = 2, _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'dp_worker.py'), str(tmp_path), str(sparse), tables],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
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
    assert r[0]['wire'][0] <= r[0]['dense'][0], (r[0]['wire'], r[0]['dense'])   # (tiny KG, 8 relations: most are in the union)
    assert str(r[0]['form'][0]) == str(r[1]['form'][0])
