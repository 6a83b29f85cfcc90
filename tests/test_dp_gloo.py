"""CPU, world_size 2, gloo: the data-parallel mechanics of spvipes_amd.train -- parameters re-homed into one
flat buffer, gradients accumulated into one flat buffer, ONE all-reduce per step, 1/world folded into the
optimiser scale.  (The HIP kernels themselves need the GPU; tests/test_gpu_parity.py covers them.)"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from spvipes_amd.train import FlatParams


def _net():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))


def _batch(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(6, 5, generator=g), torch.randn(6, 3, generator=g)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _net()
    fp = FlatParams(net)
    dist.broadcast(fp.flat, src=0)
    x, y = _batch(rank)
    fp.zero_grad()
    ((net(x) - y) ** 2).mean().backward()
    assert all(p.grad.data_ptr() >= fp.grad.data_ptr() for p in fp.params)  # grads landed in the flat buffer
    dist.all_reduce(fp.grad, op=dist.ReduceOp.SUM)  # the step's single collective
    mean_grad = fp.grad / world
    fp.flat.add_(mean_grad, alpha=-0.1)  # any optimiser acting on the flat buffers
    # numpy on the queue: a torch tensor is shared by file descriptor THROUGH the producer, which may have exited by the
    # time the parent unpickles it (FileNotFoundError / ConnectionResetError one run in four)
    q.put((rank, mean_grad.numpy().copy(), fp.flat.detach().numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_allreduce_equals_mean_of_rank_gradients():
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    # single-process reference: mean of the two ranks' gradients on the same two minibatches
    grads = []
    for r in range(world):
        net = _net()
        x, y = _batch(r)
        ((net(x) - y) ** 2).mean().backward()
        grads.append(torch.cat([torch.nn.functional.pad(p.grad.flatten(), (0, (-p.numel()) % 4)) for p in net.parameters()]))
    want = (grads[0] + grads[1]) / world
    for rank, mean_grad, flat in got:
        torch.testing.assert_close(torch.from_numpy(mean_grad), want, rtol=1e-6, atol=1e-7)
    assert (got[0][2] == got[1][2]).all()  # replicas stay bit-identical


def test_flat_params_are_views_and_survive_load_state_dict():
    net = _net()
    ref = {k: v.clone() for k, v in net.state_dict().items()}
    fp = FlatParams(net)
    for k, v in net.state_dict().items():
        torch.testing.assert_close(v, ref[k])
    net.load_state_dict({k: v + 1 for k, v in ref.items()})
    assert abs(float(fp.flat[0]) - float(ref["0.weight"].flatten()[0]) - 1) < 1e-6  # still the same storage


def test_late_parameters_form_the_second_bucket():
    net = _net()
    ref = {k: v.clone() for k, v in net.state_dict().items()}
    fp = FlatParams(net, late=lambda name: name.startswith("0."))   # first layer last, like the encoders in train.Trainer
    assert fp.split == 24 + 4  # 2.weight (21 -> padded 24) + 2.bias (3 -> padded 4)
    assert net[2].weight.data_ptr() == fp.flat.data_ptr() and net[0].weight.data_ptr() == fp.flat.data_ptr() + 4 * fp.split
    for k, v in net.state_dict().items():
        torch.testing.assert_close(v, ref[k])
    x, y = _batch(0)
    fp.zero_grad()
    ((net(x) - y) ** 2).mean().backward()
    assert float(fp.grad[:fp.split].abs().max()) > 0 and float(fp.grad[fp.split:].abs().max()) > 0


# ---- collective early stopping + BatchNorm buffer sync (a rank that stops alone deadlocks its peers) ---------------------
def _es_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spvipes_amd.train import EarlyStopping, sync_float_buffers
    # rank-local validation curves that would stop the ranks in DIFFERENT epochs if each decided alone
    curves = {0: [10.0, 9.0, 9.5, 9.6, 9.7, 9.8, 9.9], 1: [10.0, 9.0, 8.0, 7.0, 7.5, 7.6, 7.7]}
    es = EarlyStopping(patience=2, min_delta=0.0, world=world)
    stopped, n_reduce = None, 0
    for ep, v in enumerate(curves[rank]):
        g = torch.ones(3) * (rank + 1)
        dist.all_reduce(g)            # the step's gradient all-reduce: a rank that left the loop alone would hang its peer here
        n_reduce += 1
        if es.should_stop(v):
            stopped = ep
            break
    bn = torch.nn.BatchNorm1d(4)
    with torch.no_grad():
        bn.running_mean.fill_(float(rank))          # ranks hold different running statistics
        bn.running_var.fill_(1.0 + 2.0 * rank)
    sync_float_buffers(bn, world)
    q.put((rank, stopped, n_reduce, bn.running_mean.numpy().copy(), bn.running_var.numpy().copy(), int(bn.num_batches_tracked)))
    dist.barrier()
    dist.destroy_process_group()


def test_early_stopping_decision_is_collective_and_bn_buffers_are_averaged():
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_es_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    # alone, rank 0 would stop at epoch 3 (best 9.0 at epoch 1) and rank 1 at epoch 5 (best 7.0 at epoch 3); on the rank mean
    # [10, 9, 8.75, 8.3, 8.6, 8.7, 8.8] both stop at epoch 5
    assert got[0][1] == got[1][1] == 5 and got[0][2] == got[1][2] == 6
    for r in range(world):
        assert (got[r][3] == 0.5).all() and got[r][3].shape == (4,) and (got[r][4] == 2.0).all() and got[r][5] == 0


def test_early_stopping_single_rank_matches_lightning_rule():
    from spvipes_amd.train import EarlyStopping
    es = EarlyStopping(patience=2, min_delta=0.5)
    out = [es.should_stop(v) for v in (10.0, 9.8, 9.4, 9.3, 9.2)]   # 9.8 is no improvement by > 0.5; 9.4 is; then two bad epochs
    assert out == [False, False, False, False, True]
