"""-m gpu: the multi-GPU launch paths of bench.py on the ONE GPU a gpurun box has (round-2 VERDICT: "no hardware evidence for N > 1").
Each test starts bench.py as a FRESH CHILD PROCESS (RCCL initialisation must not share a process with the test runner's HIP
context) at a tiny workload and checks the driver's JSON contract:
  * --force-dist: one rank initialises torch.distributed with the nccl (= RCCL) backend: init, barrier, MAX all-reduce of the timing;
  * --gpus 1 --spawn: the `python bench.py --gpus N` path -- bench.py re-launches itself through torch.distributed.run
    (spawn_ranks), rank 0 of the child prints the line;
and the gradient exchange of config C5 through the C ABI (vtgb_comm_* / vtgb_allreduce_f32) on a one-rank communicator."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(extra):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "1", "--warmup", "1", "--clips", "2", "--no-cpu-baseline", "--no-secondary",
           "--raft-clips", "2"] + extra
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                       # ONE JSON line from rank 0
    return json.loads(lines[0]), p.stderr


def _check_line(d):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["scaling"] == "weak" and d["unit"] == "clips/s" and d["higher_is_better"] is True
    assert d["value"] > 0 and abs(d["value"] - 2 * 1000.0 / d["ms_per_step"]) < 1e-2 * d["value"]        # 2 clips per step on 1 GPU
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1


def test_bench_one_rank_over_rccl():
    d, err = _run_bench(["--force-dist"])
    _check_line(d)
    assert "world=1" in err


def test_bench_spawned_through_torch_distributed_run():
    d, err = _run_bench(["--gpus", "1", "--spawn"])
    _check_line(d)


def test_allreduce_f32_through_the_c_abi_one_rank():
    """vtgb_comm_unique_id / vtgb_comm_init / vtgb_allreduce_f32 / vtgb_comm_destroy on a one-rank communicator (sum and mean
    are the identity), and FlatGradBucket's overlapped path with that communicator: gradients are views, segments leave from the
    backward hooks on the side stream, all_reduce() joins them."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from videotgb_amd import dist as vd
    dev = torch.device("cuda:0")
    comm = vd.RcclComm(dev)
    x = torch.arange(1 << 20, dtype=torch.float32, device=dev)
    y = x.clone()
    comm.all_reduce_(y, average=False)
    comm.all_reduce_(y, average=True)
    torch.cuda.synchronize()
    assert torch.equal(x, y)
    with pytest.raises(ValueError):
        comm.all_reduce_(x.half())
    net = torch.nn.Sequential(torch.nn.Linear(256, 512), torch.nn.Linear(512, 64)).to(dev)
    bucket = vd.FlatGradBucket(net.parameters(), segment_bytes=64 << 10, comm=comm)
    inp = torch.randn(8, 256, device=dev)
    net(inp).sum().backward()                       # reference gradients (accumulated into the views)
    ref = bucket.flat.clone()
    bucket.zero_()
    bucket.arm(average=True)
    net(inp).sum().backward()
    assert sum(bucket._launched) == len(bucket.segments) > 1
    bucket.all_reduce(average=True)
    torch.cuda.synchronize()
    assert torch.allclose(bucket.flat, ref, rtol=1e-6, atol=1e-6)
    assert all(p.grad.data_ptr() == bucket.flat.data_ptr() + 4 * o for p, o in zip(bucket.params, bucket.offsets))
    comm.close()
