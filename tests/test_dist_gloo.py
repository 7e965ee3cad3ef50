"""Batch sharding over ranks (world_size 2, gloo, CPU): the sharded sweep - every rank holding
N/G class-sorted series, one all-reduce of (gradient, loss) per optimiser step with the GLOBAL
class counts as divisors, then the identical update + SVD on every rank - reproduces the
unsharded oracle.  This is the algebra libmpstime_hip.so implements with RCCL
(mpst_set_dataset(..., n_global_per_class) + ncclAllReduce of the gradient buffer)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, loss, sep, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mpstime_jl_amd as mt
    from oracle import ref_numpy as R
    from tests.helpers import make_problem

    ds, W0 = make_problem(46, 5, 3, 3, 3, seed=21, balanced=False)
    full = mt.EncodedTimeSeriesSet(ds.phi, ds.label_index.astype(np.int64), ds.label_index, np.zeros((0, 0)),
                                   ds.class_distribution)
    local, gcounts = mt.Shard(rank, world).split(full)
    loc = R.EncodedSet(local.phi, local.label_index, local.class_distribution)
    Ng = int(gcounts.sum())
    opts = R.SweepOptions(chi_max=6, eta=0.05, loss_grad=loss, train_classes_separately=sep)
    lg = R.LOSS_GRADS[loss]

    def sharded_loss_grad(bt, LE, RE, data, lid, rid, train_sep=False):
        # local partial sums re-weighted to the global divisors, then one all-reduce
        if loss == "KLD":
            grad = np.zeros_like(bt)
            lsum = 0.0
            for c in range(bt.shape[1]):
                n_c = int(data.class_distribution[c])
                if n_c == 0:
                    continue
                one = R.EncodedSet(data.phi, data.label_index, np.where(np.arange(bt.shape[1]) == c, n_c, 0))
                # evaluate class c alone on this shard (TrainSeparate{true} form: divisor n_c local)
                sel = data.label_index == c
                sub = R.EncodedSet(data.phi[sel], np.zeros(n_c, dtype=np.int32), np.array([n_c]))
                LEs = [None if a is None else a[sel] for a in LE]
                REs = [None if a is None else a[sel] for a in RE]
                l_c, g_c = R.loss_grad_KLD(bt[:, c:c + 1], LEs, REs, sub, lid, rid, True)
                wgt = n_c / (gcounts[c] if train_sep else Ng)
                grad[:, c] = g_c[:, 0] * wgt
                lsum += l_c * wgt
        else:
            l_loc, g_loc = lg(bt, LE, RE, data, lid, rid, False)
            wgt = data.N / Ng
            grad, lsum = g_loc * wgt, l_loc * wgt
        buf = torch.from_numpy(np.concatenate([[lsum], grad.reshape(-1)]))
        dist.all_reduce(buf)
        return float(buf[0]), buf[1:].numpy().reshape(bt.shape)

    R.LOSS_GRADS[loss] = sharded_loss_grad
    W = [t.copy() for t in W0]
    rec = []
    R.sweep(W, loc, opts, record=rec)
    R.LOSS_GRADS[loss] = lg
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), loss=[b["loss"] for b in rec], gn=[b["grad_norm"] for b in rec],
             **{f"W{j}": t for j, t in enumerate(W)})
    dist.destroy_process_group()


@pytest.mark.parametrize("loss,sep", [("KLD", False), ("KLD", True), ("MSE", False)])
def test_sharded_sweep_equals_unsharded(tmp_path, loss, sep):
    sys.path.insert(0, ROOT)
    from oracle import ref_numpy as R
    from tests.helpers import make_problem

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), loss, sep, str(tmp_path)), nprocs=world, join=True)
    ds, W0 = make_problem(46, 5, 3, 3, 3, seed=21, balanced=False)
    opts = R.SweepOptions(chi_max=6, eta=0.05, loss_grad=loss, train_classes_separately=sep)
    W = [t.copy() for t in W0]
    rec = []
    R.sweep(W, ds, opts, record=rec)
    outs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for o in outs:
        assert np.allclose(o["loss"], [b["loss"] for b in rec], rtol=1e-11, atol=1e-13)
        assert np.allclose(o["gn"], [b["grad_norm"] for b in rec], rtol=1e-11)
    # replicas are bit-identical to each other (same reduced bits -> same update and SVD) ...
    for j in range(len(W)):
        assert np.array_equal(outs[0][f"W{j}"], outs[1][f"W{j}"])
    # ... and equal to the unsharded run up to the reassociation of the sum over series
    yo = R.contract_mps(W, ds.phi)
    ys = R.contract_mps([outs[0][f"W{j}"] for j in range(len(W))], ds.phi)
    assert np.abs(yo - ys).max() < 1e-10 * np.abs(yo).max()


def _worker_complex(rank, world, port, loss, out_dir):
    """One bond of the legacy-engine (complex) gradient, sharded: local partial sums re-weighted to the global divisors + one
    all-reduce of the interleaved (re, im) message = the unsharded gradient.  The message layout is the one the library
    all-reduces ([loss, pad, grad as (re, im) pairs], fp64)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import ref_complex as RC
    from oracle import ref_numpy as R

    data, W = RC.make_problem(38, 5, 3, 4, 2, seed=4, balanced=False)
    lid, rid = 2, 3
    rng = np.random.default_rng(9)
    N = data.phi.shape[0]
    LEp = rng.standard_normal((N, W[lid].shape[0])) + 1j * rng.standard_normal((N, W[lid].shape[0]))
    REp = rng.standard_normal((N, W[rid].shape[2])) + 1j * rng.standard_normal((N, W[rid].shape[2]))
    C = 2
    bt5 = rng.standard_normal((3, W[lid].shape[0], 3, W[rid].shape[2], C)) + 1j * rng.standard_normal((3, W[lid].shape[0], 3, W[rid].shape[2], C))
    full_loss, full_grad = RC.loss_grad(bt5, LEp, REp, data, lid, rid, loss)
    # class-even shard of the class-sorted set (what Shard.split does): every rank takes every world-th series of each class
    idx, i0 = [], 0
    counts = [int(x) for x in data.class_distribution]
    for cn in counts:
        idx.extend(range(i0 + rank, i0 + cn, world))
        i0 += cn
    idx = np.array(idx)
    lc = np.array([np.sum(data.label_index[idx] == c) for c in range(C)])
    loc = R.EncodedSet(data.phi[idx], data.label_index[idx], lc)
    l_loc, g_loc = RC.loss_grad(bt5, LEp[idx], REp[idx], loc, lid, rid, loss)
    wgt = len(idx) / N                                   # local divisor N_local -> global divisor N
    msg = np.concatenate([[l_loc * wgt, 0.0], np.ascontiguousarray(g_loc * wgt).view(np.float64).reshape(-1)])
    buf = torch.from_numpy(msg)
    dist.all_reduce(buf)
    g = buf[2:].numpy().view(np.complex128).reshape(bt5.shape)
    np.savez(os.path.join(out_dir, f"crank{rank}.npz"), loss=float(buf[0]), full_loss=float(full_loss),
             err=float(np.abs(g - full_grad).max() / np.abs(full_grad).max()))
    dist.destroy_process_group()


@pytest.mark.parametrize("loss", ["KLD", "MSE"])
def test_complex_gradient_shards_over_two_ranks(tmp_path, loss):
    port = _free_port()
    mp.spawn(_worker_complex, args=(2, port, loss, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        z = np.load(tmp_path / f"crank{r}.npz")
        assert abs(float(z["loss"]) - float(z["full_loss"])) <= 1e-12 * max(1.0, abs(float(z["full_loss"])))
        assert float(z["err"]) < 1e-13
