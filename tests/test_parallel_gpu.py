"""The N > 1 path on hardware, as far as a one-GPU box allows: a real 2-rank
job of bench.py (child torch.distributed.run process, gloo rendezvous, both ranks on the one GPU) whose
gathered energies must equal, bit for bit, what this process computes for the same geometries; and the
nccl (= RCCL) backend initialised at world size 1 with the device-tensor all_gather of
parallel.gather_results executed on it."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return env


def test_two_rank_job_gathers_the_same_energies_as_one_process():
    """bench.py --gpus 2 --backend gloo as a CHILD job: 2 ranks x 4 geometries (geometry g on rank g mod 2),
    one all_gather at the end; n_gpus == 2 in its JSON line, and every gathered energy equals bitwise the
    energy this process gets for the same geometry in a batch of the same composition."""
    import bench
    geoms_per_rank = 4
    cmd = bench.launch_command(2, 0, ["--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
                                      "--geoms", str(geoms_per_rank), "--no-cpu-baseline", "--no-transform",
                                      "--no-kupccd", "--no-berry", "--prime-seconds", "0.05"])
    res = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["geometries_per_rank"] == geoms_per_rank
    assert out["scaling"] == "weak" and out["value"] > 0
    energies = [float.fromhex(h) for h in out["gathered_energies_hex"]]
    assert len(energies) == 2 * geoms_per_rank
    # the same shards in this process: rank r of 2 owns geometries r, r + 2, ...
    from auto_oo_amd.parallel import shard_geometries
    for r in range(2):
        mine = shard_geometries(2 * geoms_per_rank, r, 2)
        pqc, batch, single, thetas = bench.build_geometries(mine)
        e = batch.energy_and_gradient(thetas)[:, 0].tolist()
        for g, eg in zip(mine, e):
            assert eg == energies[g], (g, eg, energies[g])


_NCCL_CHECK = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from auto_oo_amd.parallel import gather_results, shard_geometries
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, init_method="tcp://127.0.0.1:{port}",
                        device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
mine = shard_geometries(5, 0, 1)
local = torch.arange(15, dtype=torch.float64, device="cuda").reshape(5, 3)
full = gather_results(local, mine, 5, dist)        # the device branch: all_gather of device tensors
assert full.is_cuda and torch.equal(full, local)
t = torch.ones(4, dtype=torch.float64, device="cuda")
dist.all_reduce(t)
torch.cuda.synchronize()
assert float(t.sum()) == 4.0
dist.barrier()
dist.destroy_process_group()
print("nccl world-1 ok")
"""


def test_nccl_backend_world_size_one_gathers_device_tensors():
    """RCCL is loaded and initialised (backend "nccl" IS RCCL on ROCm) and gather_results' device branch
    runs through a real all_gather on it -- at world size 1, which is all a one-GPU box allows."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    res = subprocess.run([sys.executable, "-c", _NCCL_CHECK.format(root=ROOT, port=port)], env=_env(),
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, (res.stdout[-1000:], res.stderr[-3000:])
    assert "nccl world-1 ok" in res.stdout


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: the RCCL leg with N > 1")
def test_two_gpu_rccl_job_matches_one_process():
    """The driver's own N = 2 command (`bench.py --gpus 2`, backend nccl = RCCL over xGMI, one rank per
    GPU) as a CHILD job -- never an exec of this process: n_gpus == 2, torch.distributed reports nccl with
    two ranks, the energies that crossed the all_gather equal bitwise what this process computes for the
    same shards, and the strong leg of the Berry-phase loop ran on both ranks."""
    import bench
    geoms_per_rank = 4
    cmd = bench.launch_command(2, 0, ["--gpus", "2", "--backend", "nccl", "--steps", "2", "--warmup", "1",
                                      "--geoms", str(geoms_per_rank), "--no-cpu-baseline", "--no-transform",
                                      "--no-kupccd", "--berry-geoms", "4", "--prime-seconds", "0.05"])
    res = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2
    assert out["config"]["dist_backend"] == "nccl" and out["config"]["dist_world_size"] == 2
    energies = [float.fromhex(h) for h in out["gathered_energies_hex"]]
    assert len(energies) == 2 * geoms_per_rank
    from auto_oo_amd.parallel import shard_geometries
    for r in range(2):
        mine = shard_geometries(2 * geoms_per_rank, r, 2)
        pqc, batch, single, thetas = bench.build_geometries(mine)
        e = batch.energy_and_gradient(thetas)[:, 0].tolist()
        for g, eg in zip(mine, e):
            assert eg == energies[g], (g, eg, energies[g])
    strong = out["berry_loop"]["strong"]
    assert strong["geometries"] == 4 and strong["lockstep"]["geometries_per_s"] > 0
    assert strong["lockstep"]["max_abs_energy_difference_vs_sequential"] < 1e-9
