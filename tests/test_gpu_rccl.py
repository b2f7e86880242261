"""RCCL on the one GPU this box has: the collectives of the path (all_gather_into_tensor of the exchange buffer,
the 16-byte MAX all-reduce of convert.py) executed on the "nccl" backend with world_size 1 (DISSC_FORCE_DIST=1), held
to the run without a process group byte for byte.  N > 1 over xGMI stays the driver's multi-GPU run; what this
removes is the "has never executed" risk of the NCCL branch: device_id= initialisation, device-side gather into a
fresh tensor, the unpack of the gathered 2-D buffer, barrier + destroy."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DISSC_FORCE_DIST",
                                                            "DISSC_DIST_BACKEND", "DISSC_BENCH_BACKEND")}
    env.update(kw)
    return env


def test_exchange_buffer_device_pack_equals_host_pack_and_survives_rccl():
    """dissc_pack_rows (device) writes the same bytes as the host rehearsal of the layout; an all-gather of it on
    RCCL (one rank) returns it unchanged; unpack gives back every waveform."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.distributed as dist
    from dissc_amd import harness
    rs = np.random.RandomState(0)
    lens = [0, 1, 3, 4, 5, 4095, 4096, 4097, 160000, 9001, 12288]
    waves = [torch.from_numpy(rs.standard_normal(n).astype(np.float32)) for n in lens]
    ids = list(range(100, 100 + len(lens)))
    data_cap = sum((n + 3) // 4 * 4 for n in lens) + 64
    host = harness.pack_waves(waves, ids, len(lens) + 2, data_cap, "cpu")
    used = harness.HDR + harness.ENT * (len(lens) + 2) + data_cap - 64
    # batches of several rows with a padded row stride, like the generator's output
    st = harness.WaveStore(DEV)
    order = [[0, 1, 2], [3, 4, 5, 6], [7], [8, 9, 10]]
    for grp in order:
        ld = max(lens[k] for k in grp) + 8
        w = torch.full((len(grp), ld), float("nan"), device=DEV)
        for r, k in enumerate(grp):
            w[r, :lens[k]] = waves[k].to(DEV)
        st.add(w, [lens[k] for k in grp], [ids[k] for k in grp])
    dev = st.pack(len(lens) + 2, data_cap)
    assert torch.equal(dev[:used].cpu().view(torch.int32), host[:used].view(torch.int32))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        out = harness.gather_store(st, len(lens) + 2, data_cap, 0, 1, dist)
        n_cap, d_cap = harness.agree_geometry(7, 12345, 1, DEV, dist)
    finally:
        dist.destroy_process_group()
    assert (n_cap, d_cap) == (7, 12345)
    assert sorted(out) == ids
    for k, j in enumerate(ids):
        np.testing.assert_array_equal(out[j], waves[k].numpy())


def test_sr_inference_on_rccl_world_size_one_matches_no_process_group(tmp_path):
    import synthdata as synth
    td = str(tmp_path)
    os.makedirs(f"{td}/ckpt")
    os.makedirs(f"{td}/meta")
    import shutil
    shutil.copy(os.path.join(ROOT, "tests", "golden", "vctk_id_to_spkr.pkl"), f"{td}/meta/id_to_spkr.pkl")
    cfg = dict(synth.VCTK_CONFIG, input_training_file=f"{td}/meta/train.txt", f0_normalize=False, f0_stats=None)
    json.dump(cfg, open(f"{td}/ckpt/config.json", "w"))
    torch.save({"generator": synth.synth_generator_state_dict(seed=0)}, f"{td}/ckpt/g_00000001")
    rs = np.random.RandomState(4)
    with open(f"{td}/man.txt", "w") as f:
        for u in range(48):
            T = int(rs.randint(20, 260)) if u else 0  # one empty `units` line rides along
            code, f0, _, _ = synth.synth_generator_inputs(1, max(T, 1), seed=900 + u)
            f.write(json.dumps({"units": code[0, :T].tolist(), "f0": [float(v) for v in f0[0, 0, :T]],
                                "audio": f"p{225 + u % 7}_{u:03d}.wav"}) + "\n")
    args = [os.path.join(ROOT, "sr", "inference.py"), "--input_code_file", f"{td}/man.txt", "--data_path", f"{td}/nowav",
            "--checkpoint_file", f"{td}/ckpt/", "--vc", "--target-speakers", "p231", "p225", "--unseen_speaker",
            "--id_to_spkr", f"{td}/meta/id_to_spkr.pkl", "-n", "-1"]
    for name, env in (("plain", _env()), ("rccl", _env(DISSC_FORCE_DIST="1", MASTER_PORT=str(_free_port())))):
        r = subprocess.run([sys.executable] + args + ["--output_dir", f"{td}/{name}"], env=env, capture_output=True,
                           text=True, timeout=900, cwd=td)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    files = sorted(os.listdir(f"{td}/plain"))
    assert len(files) == 96 and sorted(os.listdir(f"{td}/rccl")) == files
    for fn in files:
        assert open(f"{td}/plain/{fn}", "rb").read() == open(f"{td}/rccl/{fn}", "rb").read(), fn


def test_bench_on_rccl_world_size_one():
    """bench.py with the process group forced: the weak-scaling step's all-gather and the strong leg's exchange run on
    RCCL; the line names every collective."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-split-bf16", "--no-pipeline", "--no-d2h"],
                       env=_env(DISSC_FORCE_DIST="1", MASTER_PORT=str(_free_port())), capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["n_gpus"] == 1 and "nccl" in j["config"]["collective"] and j["value"] > 200
    st = j["strong"]
    # one all-gather per round; with the delivery thread a run of this size is cut into <= 4 rounds
    assert st["jobs"] == 1024 and 1 <= st["exchange"]["collectives"] == st["exchange"]["rounds"] <= 4 and st["value"] > 200
    assert st["exchange"]["overlap"] is True
    assert st["exchange"]["sent_bytes_per_rank"] <= 1.1 * st["exchange"]["payload_bytes_this_rank"]


def test_bench_two_ranks_sharing_this_gpu():
    """bench.py --gpus 2 on the HIP path with both ranks on this box's one GPU (gloo stages the device tensors through
    the host; RCCL needs one GPU per rank): the weak-scaling step with its all-gather, and the strong-scaling leg with
    the job list really split over two ranks -- per-rank compute times, one collective, LPT imbalance close to 1."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-split-bf16", "--no-pipeline", "--no-d2h"],
                       env=_env(DISSC_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and "gloo" in j["config"]["collective"]
    st = j["strong"]
    assert st["jobs"] == 1024 and len(st["per_rank_compute_ms"]) == 2
    assert 1 <= st["exchange"]["collectives"] == st["exchange"]["rounds"] <= 4  # one all-gather per (overlapped) round
    assert st["own_rows"]["value"] > 200  # the CLIs' N > 1 default: every rank delivers what it decoded
    assert 1.0 <= st["load_imbalance"] < 1.01 and st["value"] > 200
    # each rank sends its half of the waveforms (plus table and 16-byte row padding), not a dense matrix
    assert st["exchange"]["sent_bytes_per_rank"] <= 0.55 * 4 * 16000 * st["audio_sec"]


def test_to_host_double_buffered_path_on_a_non_current_device():
    """harness._to_host above 32 MB works through two page-locked halves with events: copies and events must go to the
    TENSOR's device's stream also when another device is current.  Needs two GPUs; on the one-GPU box the
    same path is checked on the current device."""
    from dissc_amd import harness
    n = (40 << 20) // 4 + 12345  # > 32 MB: the double-buffered branch
    dev = torch.device("cuda", 1) if torch.cuda.device_count() > 1 else torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(0)
    want = torch.rand(n, generator=g)
    t = want.to(dev)
    torch.cuda.synchronize(dev)
    with torch.cuda.device(0):  # device 0 current, tensor possibly on device 1
        # some work on the tensor's stream right before, so that an unordered read would see stale staging halves
        t.mul_(2.0).mul_(0.5)
        got = harness._to_host(t)
    assert got.shape == (n,) and np.array_equal(got, want.numpy())


def _cabi_rank(rank, world, uid_path, out_path):
    """one rank of the C-ABI all-gather: its own GPU, a WaveComm made from the id rank 0 wrote, one gather_store round"""
    import time
    import torch as th
    from dissc_amd import collective, harness
    dev = f"cuda:{rank}"
    th.cuda.set_device(rank)
    if rank == 0:
        uid = collective.unique_id()
        with open(uid_path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(uid_path + ".tmp", uid_path)
    else:
        for _ in range(600):
            if os.path.exists(uid_path):
                break
            time.sleep(0.1)
        uid = open(uid_path, "rb").read()
    comm = collective.WaveComm(uid, world, rank, device=dev)
    rs = np.random.RandomState(100 + rank)
    lens = [4097 + 13 * rank, 1, 160000 - 7 * rank, 0, 12288]
    ids = [1000 * rank + k for k in range(len(lens))]
    st = harness.WaveStore(dev)
    w = th.full((len(lens), max(lens) + 8), float("nan"), device=dev)
    for r, n in enumerate(lens):
        w[r, :n] = th.from_numpy(rs.standard_normal(n).astype(np.float32)).to(dev)
    st.add(w, lens, ids)
    n_cap, d_cap = len(lens) + 1, 180000 + 4 * 4097 + 64
    out = harness.gather_store(st, n_cap, d_cap, rank, world, comm, unpack_ranks=None)
    comm.destroy()
    np.savez(out_path, **{str(k): v for k, v in out.items()})


def _cabi_expected(world):
    want = {}
    for rank in range(world):
        rs = np.random.RandomState(100 + rank)
        for k, n in enumerate([4097 + 13 * rank, 1, 160000 - 7 * rank, 0, 12288]):
            want[1000 * rank + k] = rs.standard_normal(n).astype(np.float32)
    return want


def test_cabi_allgather_world_size_one_equals_torch_distributed(tmp_path):
    """dissc_comm_* + dissc_allgather_waves on a world-size-1 RCCL communicator made through the C ABI: the gathered buffer is
    byte-identical to torch.distributed's all_gather_into_tensor of the same packed buffer (the CLIs' default route), in place
    and out of place, and a whole gather_store round through the WaveComm returns every waveform."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.distributed as dist
    from dissc_amd import collective, harness
    _cabi_rank(0, 1, str(tmp_path / "uid"), str(tmp_path / "r0.npz"))
    got = np.load(str(tmp_path / "r0.npz"))
    want = _cabi_expected(1)
    assert sorted(int(k) for k in got.files) == sorted(want)
    for k, v in want.items():
        np.testing.assert_array_equal(got[str(k)], v)
    # the raw collective against torch.distributed on the same bytes
    buf = torch.randn(300001, device=DEV)
    comm = collective.WaveComm(collective.unique_id(), 1, 0, device=DEV)
    out_c = torch.empty_like(buf)
    side = torch.cuda.Stream(device=DEV)
    side.wait_stream(torch.cuda.current_stream())
    comm.all_gather_into_tensor(out_c, buf, async_op=True, stream=side).wait()
    inplace = buf.clone()
    comm.all_gather_into_tensor(inplace, inplace)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        out_t = torch.empty_like(buf)
        dist.all_gather_into_tensor(out_t, buf)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    comm.destroy()
    assert torch.equal(out_c.view(torch.int32), out_t.view(torch.int32)) and torch.equal(inplace, buf)
    with pytest.raises(Exception):
        comm.all_gather_into_tensor(out_c, buf)  # destroyed


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank: auto-enables on a box with >= 2 GPUs")
def test_cabi_allgather_two_ranks_over_xgmi(tmp_path):
    """two processes, one GPU each, the communicator and the all-gather through the C ABI only (no torch.distributed)"""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_rccl as t; "
            "t._cabi_rank(int(sys.argv[1]), 2, sys.argv[2], sys.argv[3])") % (ROOT, os.path.join(ROOT, "tests"))
    uid = str(tmp_path / "uid")
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), uid, str(tmp_path / f"r{r}.npz")], env=_env(), cwd=ROOT)
             for r in range(2)]
    assert [p.wait(timeout=600) for p in procs] == [0, 0]
    want = _cabi_expected(2)
    for r in range(2):
        got = np.load(str(tmp_path / f"r{r}.npz"))
        assert sorted(int(k) for k in got.files) == sorted(want)
        for k, v in want.items():
            np.testing.assert_array_equal(got[str(k)], v)
