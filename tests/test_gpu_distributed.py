"""2-rank run == 1-rank run (sharded tables + data-parallel towers), on the one-GPU box."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('model,shard,arch', [('DeepFM', 'table', 'tiny'), ('DCN', 'table', 'tiny'),
                                              ('PNN', 'table', 'tiny'), ('DeepFM', 'row', 'tiny'),
                                              ('DCN', 'row', 'tiny'), ('PNN', 'row', 'tiny'),
                                              ('DeepFM', 'table', 'default'), ('DeepFM', 'row', 'default'),
                                              ('DCN', 'table', 'default'), ('DeepFM', 'table', 'tiny direct'),
                                              ('DCN', 'row', 'default direct'), ('DeepFM', 'table', 'tiny nobuf'),
                                              ('DCN', 'row', 'tiny nobuf')])
def test_two_ranks_match_one_rank(sess, model, shard, arch):
    """'... direct': the towers' gradients summed by distributed.DirectAllReduce (two all-to-alls and a fixed-order sum)
    instead of the transport's all-reduce; '... nobuf': adam_mode 'tf1' as the row-wise pass plus the pass over the rows
    without a gradient (table_grads=False) on every rank's shard."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
           '--master-addr', '127.0.0.1', '--master-port', '29731', os.path.join(ROOT, 'tests', 'dist_worker.py'), model, shard] + arch.split()
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'DIST_OK %s %s' % (model, shard) in r.stdout


def test_rccl_transport_one_rank(sess):
    """distributed.TorchComm over the nccl (= RCCL) backend: one rank, both sharded layouts, against the
    unsharded front end (a one-GPU box cannot hold two RCCL ranks; the 2-rank cases above use the host-staged
    transport)."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1',
           '--master-addr', '127.0.0.1', '--master-port', '29733', os.path.join(ROOT, 'tests', 'rccl_worker.py')]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'RCCL_OK backend=nccl' in r.stdout


def test_library_collectives_one_rank(sess):
    """ctr_comm_* / ctr_all2all(v) / ctr_allreduce_sum_f32 (RCCL behind the C ABI) and distributed.CtrComm: one rank,
    raw calls and both sharded layouts against the unsharded front end, in a process of its own."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'comm_abi_worker.py')], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'CTR_COMM_OK' in r.stdout
