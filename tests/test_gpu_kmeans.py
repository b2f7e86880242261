"""The integer step of the path in isolation: dissc_kmeans_assign (HIP) == oracle_kmeans_assign_f32 (C restatement of the same
bit-exact specification) == sklearn KMeans.predict on the committed golden features -- np.array_equal, exact ties included.
Reference: data/encode.py:21-22 (textless KMeansQuantizer -> sklearn predict); SURVEY section 8(b) export list."""
import ctypes
import os

import numpy as np
import pytest
import torch

from test_oracle_golden import kmeans_f32, kmeans_tie_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def coracle():
    import __graft_entry__ as ge
    return ctypes.CDLL(ge.build_oracle())


@pytest.fixture(scope="module")
def L():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from dissc_amd import _lib
    return _lib


def hip_assign(L, dense, centers, cnorm=None):
    x = torch.from_numpy(np.ascontiguousarray(dense, dtype=np.float32)).cuda()
    c = torch.from_numpy(np.ascontiguousarray(centers, dtype=np.float32)).cuda()
    cn = torch.from_numpy(cnorm).cuda() if cnorm is not None else None
    u = torch.full((x.shape[0],), -7, dtype=torch.int64, device="cuda:0")
    L.check(L.lib.dissc_kmeans_assign(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(c.data_ptr()),
                                      ctypes.c_void_p(cn.data_ptr()) if cn is not None else None, x.shape[0], c.shape[0], c.shape[1],
                                      ctypes.c_void_p(u.data_ptr()), None), "dissc_kmeans_assign")
    torch.cuda.synchronize()
    return u.cpu().numpy()


def test_kmeans_assign_is_sklearn_predict_on_the_golden_features(L, coracle, golden_dir):
    import synthdata as synth
    g = np.load(os.path.join(golden_dir, "hubert.npz"))
    centers = synth.synth_kmeans_centers().numpy()
    for n in (400, 719, 4000, 16000, 32000):
        u = hip_assign(L, g[f"n{n}/dense"], centers)
        np.testing.assert_array_equal(u, g[f"n{n}/units"])                      # sklearn.KMeans.predict (make_golden.py)
        np.testing.assert_array_equal(u, kmeans_f32(coracle, g[f"n{n}/dense"], centers))


def test_kmeans_assign_exact_ties_duplicates_nan(L, coracle):
    x, c = kmeans_tie_cases()
    u = hip_assign(L, x, c)
    np.testing.assert_array_equal(u, kmeans_f32(coracle, x, c))
    assert u[0] == 5 and u[1] == 5 and u[2] == 60 and u[3] == 60 and u[4] == 30 and u[5] == 30 and u[20] == 0


@pytest.mark.parametrize("T,K,D", [(3000, 100, 768), (517, 50, 768), (300, 200, 768), (130, 500, 768), (65, 7, 30), (1, 1, 1),
                                   (200, 129, 64)])
def test_kmeans_assign_random_shapes_bit_exact(L, coracle, T, K, D):
    rs = np.random.RandomState(T + K)
    c = rs.standard_normal((K, D)).astype(np.float32)
    x = (c[rs.randint(0, K, T)] + 0.9 * rs.standard_normal((T, D))).astype(np.float32)  # clustered: small margins
    want = kmeans_f32(coracle, x, c)
    np.testing.assert_array_equal(hip_assign(L, x, c), want)
    cn = np.zeros(K, np.float32)  # a caller-supplied cnorm is used as given
    coracle.oracle_kmeans_cnorm_f32(c.ctypes.data_as(ctypes.c_void_p), K, D, cn.ctypes.data_as(ctypes.c_void_p))
    np.testing.assert_array_equal(hip_assign(L, x, c, cn + 1.0), kmeans_f32(coracle, x, c, cn + 1.0))


def test_encoder_units_are_kmeans_assign_of_its_own_features(L, coracle):
    """dissc_hubert_forward's integer step IS this entry: units == dissc_kmeans_assign(dense) == the C oracle on the same dense"""
    from dissc_amd.hubert import HubertEncoder
    import synthdata as synth
    centers = synth.synth_kmeans_centers()
    enc = HubertEncoder(synth.synth_hubert_state_dict(6), centers, n_layers=6).to("cuda:0")
    ns = [48000, 20000, 7777]
    wav = torch.zeros(len(ns), max(ns))
    for i, n in enumerate(ns):
        wav[i, :n] = torch.from_numpy(synth.synth_waveform(n, seed=n))
    out = enc(wav, n_samples=torch.tensor(ns))
    for i in range(len(ns)):
        T = int(out["frames"][i])
        dense = out["dense"][i][:T].cpu().numpy()
        units = out["units"][i][:T].cpu().numpy()
        np.testing.assert_array_equal(units, hip_assign(L, dense, centers.numpy()))
        np.testing.assert_array_equal(units, kmeans_f32(coracle, dense, centers.numpy()))
