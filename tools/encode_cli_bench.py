#!/usr/bin/env python
"""Wall time of data/encode.py on a directory of synthetic wavs (python tools/encode_cli_bench.py [n_files] [seconds]):
what a user of the CLI sees -- file reads, batching, HuBERT + k-means, YAAPT, JSON lines -- after a warm-up run."""
import importlib.util, json, os, shutil, sys, tempfile, time
import numpy as np, torch
from scipy.io import wavfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synthdata as synth
from test_yaapt import voiced
n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 128
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
td = tempfile.mkdtemp()
os.makedirs(f"{td}/ckpt"); os.makedirs(f"{td}/wav")
torch.save({"model": synth.synth_hubert_state_dict(6)}, f"{td}/ckpt/hubert-base-ls960.pt")
np.save(f"{td}/ckpt/kmeans_100.npy", synth.synth_kmeans_centers().numpy())
rs = np.random.RandomState(0)
for i in range(n_files):
    n = int(seconds * 16000 * (0.6 + 0.4 * rs.rand()))
    x = voiced(np.linspace(100 + i % 50, 160 + i % 70, n)) + 0.01 * rs.standard_normal(n)
    wavfile.write(f"{td}/wav/u{i:04d}.wav", 16000, np.clip(x * 20000, -32767, 32767).astype(np.int16))
spec = importlib.util.spec_from_file_location("enc_cli", os.path.join(ROOT, "data/encode.py"))
cli = importlib.util.module_from_spec(spec); spec.loader.exec_module(cli)
res = {}
for f0 in ("yaapt", "zeros"):
    for rep in range(2):
        out = f"{td}/out_{f0}_{rep}.txt"
        torch.cuda.synchronize(); t0 = time.perf_counter()
        cli.main(["--base_dir", f"{td}/wav", "--out_file", out, "--checkpoint_dir", f"{td}/ckpt", "--f0", f0])
        torch.cuda.synchronize(); res[f0] = time.perf_counter() - t0
audio = sum(len(json.loads(l)["units"]) for l in open(out)) * 0.02
print(json.dumps({"files": n_files, "audio_sec": round(audio, 1), "wall_s_yaapt": round(res["yaapt"], 3),
                  "wall_s_zeros": round(res["zeros"], 3), "x_realtime_yaapt": round(audio / res["yaapt"], 1)}))
shutil.rmtree(td)
