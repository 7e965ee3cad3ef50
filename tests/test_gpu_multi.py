"""The sharded sweep of libmpstime_hip.so itself on two ranks (one process per GPU, RCCL all-reduce of the bond gradient
once per optimiser step) against the same sweep on one GPU.  Needs two GPUs: skipped on the 1-GPU boxes of this pool."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["MPST_ROOT"])
import numpy as np
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dist.init_process_group("nccl", device_id=torch.device("cuda", rank))
import mpstime_jl_amd as mt
from oracle import ref_numpy as R
from tests.helpers import make_problem
loss, sep = os.environ["MPST_LOSS"], os.environ["MPST_SEP"] == "1"
ds, W0 = make_problem(150, 8, 4, 4, 3, seed=21, balanced=False)
full = mt.EncodedTimeSeriesSet(ds.phi, ds.label_index.astype(np.int64), ds.label_index, np.zeros((0, 0)), ds.class_distribution)
sh = mt.Shard(rank, world)
local, gcounts = sh.split(full)
eng = mt.SweepEngine(rank)
eng.set_options(chi_max=10, eta=0.05, loss=loss, train_classes_separately=sep)
sh.attach(eng)
eng.set_dataset(0, local.phi, local.label_index, 3, gcounts)
eng.set_mps(W0)
eng.build_caches()
for _ in range(2):
    eng.sweep()
ev = eng.eval(0)
np.savez(os.path.join(os.environ["MPST_OUT"], f"rank{rank}.npz"), mse=ev[0], kld=ev[1], acc=ev[2], conf=ev[3],
         **{f"W{j}": t for j, t in enumerate(eng.get_mps())})
eng.close()
dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("loss,sep", [("KLD", False), ("KLD", True), ("MSE", False)])
def test_two_rank_sweep_equals_single_gpu(tmp_path, loss, sep):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import mpstime_jl_amd as mt
    from oracle import ref_numpy as R
    from tests.helpers import make_problem
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MPST_ROOT=ROOT, MPST_OUT=str(tmp_path), MPST_LOSS=loss, MPST_SEP="1" if sep else "0",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                    "--master-port", str(_free_port()), str(script)], check=True, env=env, timeout=600)
    outs = [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]
    ds, W0 = make_problem(150, 8, 4, 4, 3, seed=21, balanced=False)
    eng = mt.SweepEngine(0)
    try:
        eng.set_options(chi_max=10, eta=0.05, loss=loss, train_classes_separately=sep)
        eng.set_dataset(0, ds.phi, ds.label_index, 3)
        eng.set_mps(W0)
        eng.build_caches()
        for _ in range(2):
            eng.sweep()
        ev = eng.eval(0)
        W1 = eng.get_mps()
    finally:
        eng.close()
    T = len(W0)
    # replicas are bit-identical: every rank applied the same update and SVD to the same all-reduced bits
    for j in range(T):
        assert np.array_equal(outs[0][f"W{j}"], outs[1][f"W{j}"])
    # evaluation sums over shards (all-reduced inside mpst_eval): same numbers on both ranks, equal to the single-GPU run
    assert outs[0]["kld"] == outs[1]["kld"] and np.array_equal(outs[0]["conf"], outs[1]["conf"])
    assert abs(float(outs[0]["kld"]) - ev[1]) <= 1e-8 * max(1.0, abs(ev[1]))
    assert np.array_equal(outs[0]["conf"], ev[3])
    yo = R.contract_mps(W1, ds.phi)
    ys = R.contract_mps([outs[0][f"W{j}"] for j in range(T)], ds.phi)
    assert np.abs(yo - ys).max() < 1e-8 * np.abs(yo).max()


ONESHOT_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["MPST_ROOT"])
import numpy as np
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
import mpstime_jl_amd as mt
from tests.helpers import make_problem
loss = os.environ["MPST_LOSS"]
dev = rank % max(torch.cuda.device_count(), 1)
ds, W0 = make_problem(150, 8, 4, 4, 3, seed=21, balanced=False)
full = mt.EncodedTimeSeriesSet(ds.phi, ds.label_index.astype(np.int64), ds.label_index, np.zeros((0, 0)), ds.class_distribution)
sh = mt.Shard(rank, world, rccl=False, oneshot=True)
local, gcounts = sh.split(full)
eng = mt.SweepEngine(dev)
eng.set_options(chi_max=10, eta=0.05, loss=loss, update_iters=2, track_cost=True)
eng.set_dataset(0, local.phi, local.label_index, 3, gcounts)
eng.set_mps(W0)
sh.attach_oneshot(eng)
eng.build_caches()
for _ in range(2):
    eng.sweep()
ev = eng.eval(0)
np.savez(os.path.join(os.environ["MPST_OUT"], f"rank{rank}.npz"), mse=ev[0], kld=ev[1], acc=ev[2], conf=ev[3], trace=eng.loss_trace(),
         **{f"W{j}": t for j, t in enumerate(eng.get_mps())})
dist.barrier()
eng.close()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("loss", ["KLD", "MSE"])
def test_oneshot_allreduce_ranks_sharing_the_gpus(tmp_path, loss, world):
    """The one-shot direct-write all-reduce (peer-mapped inboxes over hipIpc, flags, rank-order sum) without RCCL: `world`
    processes, rank r on GPU r mod #GPUs - on a 1-GPU box all ranks share the device, which exercises the whole
    protocol (IPC export / attach, push, flags, parity double-buffering, bounded spins) except the xGMI hop itself."""
    import mpstime_jl_amd as mt
    from oracle import ref_numpy as R
    from tests.helpers import make_problem
    script = tmp_path / "worker.py"
    script.write_text(ONESHOT_WORKER)
    env = dict(os.environ, MPST_ROOT=ROOT, MPST_OUT=str(tmp_path), MPST_LOSS=loss,
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                    "--master-port", str(_free_port()), str(script)], check=True, env=env, timeout=600)
    outs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    ds, W0 = make_problem(150, 8, 4, 4, 3, seed=21, balanced=False)
    eng = mt.SweepEngine(0)
    try:
        eng.set_options(chi_max=10, eta=0.05, loss=loss, update_iters=2, track_cost=True)
        eng.set_dataset(0, ds.phi, ds.label_index, 3)
        eng.set_mps(W0)
        eng.build_caches()
        for _ in range(2):
            eng.sweep()
        ev = eng.eval(0)
        W1 = eng.get_mps()
        trace1 = eng.loss_trace()
    finally:
        eng.close()
    T = len(W0)
    # track_cost on a sharded fit: every entry of the trace - the losses before each optimiser step AND the one at the
    # updated bond tensor - is the global loss, the same on every rank and equal to the single-rank run's
    for r in range(world):
        assert np.array_equal(outs[0]["trace"], outs[r]["trace"])
    assert np.abs(outs[0]["trace"] - trace1).max() <= 1e-8 * max(1.0, np.abs(trace1).max())
    for r in range(1, world):
        for j in range(T):
            assert np.array_equal(outs[0][f"W{j}"], outs[r][f"W{j}"])           # bit-identical replicas
        assert outs[0]["kld"] == outs[r]["kld"] and np.array_equal(outs[0]["conf"], outs[r]["conf"])
    assert abs(float(outs[0]["kld"]) - ev[1]) <= 1e-8 * max(1.0, abs(ev[1]))
    assert np.array_equal(outs[0]["conf"], ev[3])
    yo = R.contract_mps(W1, ds.phi)
    ys = R.contract_mps([outs[0][f"W{j}"] for j in range(T)], ds.phi)
    assert np.abs(yo - ys).max() < 1e-8 * np.abs(yo).max()


IMPUTE_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["MPST_ROOT"])
import numpy as np
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
import mpstime_jl_amd as mt
z = np.load(os.environ["MPST_PROBLEM"])
T = z["X"].shape[1]
W = [z[f"W{j}"] for j in range(T)]
opts = mt.MPSOptions(d=int(z["d"]), chi_max=int(z["chi"]), verbosity=-1)
td = mt.EncodedTimeSeriesSet(None, z["ytr"], z["ytr"].astype(np.int32), z["Xtr"], np.bincount(z["ytr"]))
imp = mt.init_imputation_problem(mt.TrainedMPS(W, opts, td), z["X"], z["y"], dx=1e-3, verbosity=0)
sh = mt.Shard(rank, world, rccl=False)
dev = rank % max(torch.cuda.device_count(), 1)
ts, err = mt.impute_dataset(imp, z["mask"], "median", shard=sh, device=dev)
np.savez(os.path.join(os.environ["MPST_OUT"], f"imp{rank}.npz"), ts=ts, err=err)
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_imputation_equals_single_process(tmp_path, world):
    """configs[4]'s multi-GPU shape: instances are independent, so the ranks impute disjoint rows with no collective on
    the data path and gather the results on the host; rank r on GPU r mod #GPUs (all on one device on this pool)."""
    import mpstime_jl_amd as mt
    from oracle import ref_numpy as R
    rng = np.random.default_rng(8)
    T, d, chi, C, Ntr, N = 12, 4, 6, 2, 20, 11          # N not a multiple of the world size: ragged shards
    W = R.random_mps(T, d, chi, C, rng)
    ytr = np.sort(rng.integers(0, C, Ntr))
    Xtr = rng.normal(size=(Ntr, T))
    y = rng.integers(0, C, N)
    X = rng.normal(size=(N, T))
    mask = rng.uniform(size=(N, T)) < 0.4
    prob = tmp_path / "problem.npz"
    np.savez(prob, X=X, y=y, Xtr=Xtr, ytr=ytr, mask=mask, d=d, chi=chi, **{f"W{j}": t for j, t in enumerate(W)})
    script = tmp_path / "worker.py"
    script.write_text(IMPUTE_WORKER)
    env = dict(os.environ, MPST_ROOT=ROOT, MPST_OUT=str(tmp_path), MPST_PROBLEM=str(prob))
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                    "--master-port", str(_free_port()), str(script)], check=True, env=env, timeout=600)
    opts = mt.MPSOptions(d=d, chi_max=chi, verbosity=-1)
    td = mt.EncodedTimeSeriesSet(None, ytr, ytr.astype(np.int32), Xtr, np.bincount(ytr))
    imp = mt.init_imputation_problem(mt.TrainedMPS(W, opts, td), X, y, dx=1e-3, verbosity=0)
    ts, err = mt.impute_dataset(imp, mask, "median")
    for r in range(world):
        o = np.load(tmp_path / f"imp{r}.npz")
        assert np.array_equal(o["ts"], ts) and np.array_equal(o["err"], err)


RCCL1_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["MPST_ROOT"])
import ctypes as C
import numpy as np
import torch                                    # loads torch's librccl: the copy libmpstime_hip.so binds at run time
import mpstime_jl_amd as mt
from tests.helpers import make_problem
big = os.environ["MPST_CASE"] == "big"
ds, W0 = make_problem(150, 8, 4, 4 if not big else 12, 3, seed=21, balanced=False)
res = {}
for forced in (0, 1):
    eng = mt.SweepEngine(0)
    eng.set_options(chi_max=10 if not big else 40, eta=0.05, update_iters=2, track_cost=True)
    if forced:
        os.environ["MPST_FORCE_COLLECTIVE"] = "1"
        lib = mt._lib.load()
        uid = (C.c_uint8 * 128)()
        assert lib.mpst_comm_unique_id(uid) == 0, lib.mpst_last_error(None)
        eng._chk(lib.mpst_comm_init(eng.ctx, uid, 1, 0))
    eng.set_dataset(0, ds.phi, ds.label_index, 3)
    eng.set_mps(W0)
    eng.build_caches()
    for _ in range(2):
        eng.sweep()
    res[forced] = dict(ev=eng.eval(0), W=eng.get_mps(), trace=eng.loss_trace(), info=eng.info(), prof=None)
    if forced:
        eng.set_profile(1 << 8)
        eng.sweep()
        res[forced]["prof"] = eng.get_profile()["allreduce"]
    eng.close()
assert res[1]["info"]["graph"] is False and res[0]["info"]["graph"] is (not big)
assert res[1]["prof"][1] == 2 * 7 * 2, res[1]["prof"]            # one ncclAllReduce per optimiser step: 14 bonds x 2 iterations
from oracle import ref_numpy as R
y0, y1 = R.contract_mps(res[0]["W"], ds.phi), R.contract_mps(res[1]["W"], ds.phi)
assert np.abs(y0 - y1).max() < 1e-9 * np.abs(y0).max(), np.abs(y0 - y1).max()
assert np.abs(res[0]["trace"] - res[1]["trace"]).max() < 1e-9 * max(1.0, np.abs(res[0]["trace"]).max())
assert abs(res[0]["ev"][1] - res[1]["ev"][1]) < 1e-9 * max(1.0, abs(res[0]["ev"][1])) and np.array_equal(res[0]["ev"][3], res[1]["ev"][3])
print("RCCL leg ok:", mt.comm_library())
"""


@pytest.mark.parametrize("case", ["fused", "big"])
def test_rccl_leg_runs_with_a_single_rank_communicator(tmp_path, case):
    """The RCCL branch of the launch chain on a 1-GPU box: a communicator of ONE rank and MPST_FORCE_COLLECTIVE=1 send every
    optimiser step through ncclAllReduce on the engine's stream (and the sweep through the plain-stream, loss-in-the-message,
    norm-from-the-summed-gradient chain the sharded fit uses).  Same results as the single-rank chain to 1e-9; the number of
    all-reduces is checked.  RCCL refuses two ranks on one device, so more than one rank needs more than one GPU
    (test_two_rank_sweep_equals_single_gpu)."""
    script = tmp_path / "worker.py"
    script.write_text(RCCL1_WORKER)
    env = dict(os.environ, MPST_ROOT=ROOT, MPST_CASE=case, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("MPST_FORCE_COLLECTIVE", None)
    r = subprocess.run([sys.executable, str(script)], env=env, timeout=600, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "RCCL leg ok" in r.stdout


TYPED_ONESHOT_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["MPST_ROOT"])
import numpy as np
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
import mpstime_jl_amd as mt
from oracle import ref_complex as RC
dt = np.dtype(os.environ["MPST_DTYPE"])
dev = rank % max(torch.cuda.device_count(), 1)
ds, W0 = RC.make_problem(150, 8, 4, 4, 3, seed=21, dtype=dt, balanced=False)
full = mt.EncodedTimeSeriesSet(ds.phi, ds.label_index.astype(np.int64), ds.label_index, np.zeros((0, 0)), ds.class_distribution)
sh = mt.Shard(rank, world, rccl=False, oneshot=True)
local, gcounts = sh.split(full)
eng = mt.SweepEngine(dev)
eng.set_options(chi_max=10, eta=0.05)
eng.set_dataset(0, local.phi, local.label_index, 3, gcounts)
eng.set_mps(W0)
sh.attach_oneshot(eng)
eng.build_caches()
for _ in range(2):
    eng.sweep()
ev = eng.eval(0)
np.savez(os.path.join(os.environ["MPST_OUT"], f"rank{rank}.npz"), kld=ev[1], conf=ev[3], **{f"W{j}": t for j, t in enumerate(eng.get_mps())})
dist.barrier()
eng.close()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("dtype", ["complex64", "float32"])
def test_sharded_typed_sweep_over_the_oneshot_allreduce(tmp_path, dtype):
    """The element-typed sweep sharded over two ranks (sharing the GPU on this pool): the all-reduce message is the fp64 gradient
    buffer - interleaved (re, im) pairs for a complex model, twice the real length - through the one-shot kernel; replicas are
    bit-identical and agree with the single-rank fit."""
    import mpstime_jl_amd as mt
    from oracle import ref_complex as RC
    script = tmp_path / "worker.py"
    script.write_text(TYPED_ONESHOT_WORKER)
    env = dict(os.environ, MPST_ROOT=ROOT, MPST_OUT=str(tmp_path), MPST_DTYPE=dtype, MPST_AR_WG="8",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                    "--master-port", str(_free_port()), str(script)], check=True, env=env, timeout=600)
    outs = [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]
    dt = np.dtype(dtype)
    ds, W0 = RC.make_problem(150, 8, 4, 4, 3, seed=21, dtype=dt, balanced=False)
    eng = mt.SweepEngine(0)
    try:
        eng.set_options(chi_max=10, eta=0.05)
        eng.set_dataset(0, ds.phi, ds.label_index, 3)
        eng.set_mps(W0)
        eng.build_caches()
        for _ in range(2):
            eng.sweep()
        ev = eng.eval(0)
        W1 = eng.get_mps()
    finally:
        eng.close()
    T = len(W0)
    for j in range(T):
        assert np.array_equal(outs[0][f"W{j}"], outs[1][f"W{j}"])
    assert outs[0]["kld"] == outs[1]["kld"] and np.array_equal(outs[0]["conf"], outs[1]["conf"])
    wide = np.complex128 if dt.kind == "c" else np.float64
    yo = RC.contract_mps([t.astype(wide) for t in W1], ds.phi.astype(wide))
    ys = RC.contract_mps([outs[0][f"W{j}"].astype(wide) for j in range(T)], ds.phi.astype(wide))
    assert np.abs(yo - ys).max() < 2e-3 * np.abs(yo).max()          # fp32 storage, two chaotic sweeps: the shards sum in another order
    assert abs(float(outs[0]["kld"]) - ev[1]) <= 2e-3 * max(1.0, abs(ev[1]))
