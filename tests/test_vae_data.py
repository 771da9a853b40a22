"""``pangaea_amd.data.Data`` and ``pangaea_amd.models.VAENET`` against vectors produced by the reference's own modules
(tests/golden/data_g4.npz, vae_g5.npz) and against the oracle's restatements; CPU-torch and GPU variants.
Tolerance for latents: 1e-5 relative to max |mu| (north-star), fp32; Data is bit-exact (fp64 division of integers)."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle
from pangaea_amd.data import Data
from pangaea_amd.loader import shuffled_batches, weighted_batches
from pangaea_amd.models.VAENET import VAENET, VaritionalAutoEncoder

from .conftest import GOLDEN

DEVICES = ["cpu", pytest.param("cuda:0", marks=pytest.mark.gpu)]


def _g5(name="vae_g5.npz"):
    g = np.load(os.path.join(GOLDEN, name))
    return g, {k[len("state/"):]: g[k] for k in g.files if k.startswith("state/")}


# the reference's own modules produced both: a [48, 40] net (small fixture) and the net its constructor builds by default --
# hidden sizes [512, 512] (VAENET.py:193), the sizes every real run has
FIXTURES = [("vae_g5.npz", [48, 40], 5), ("vae_g5b_512.npz", None, 30)]


@pytest.mark.parametrize("device", DEVICES)
def test_data_is_bit_identical_to_reference(device):
    g = np.load(os.path.join(GOLDEN, "data_g4.npz"))
    d = Data(np.array([f"bc{i}" for i in range(37)], dtype=object), g["abd_in"], g["tnf_in"], device=device)
    assert d.abd.dtype == np.float32 and d.tnf.dtype == np.float32 and d.weights.dtype == np.float64
    assert np.array_equal(d.abd, g["abd"]) and np.array_equal(d.tnf, g["tnf"]) and np.array_equal(d.weights, g["weights"])
    item = d[3]
    assert set(item) == {"abd", "tnf", "bc"} and item["bc"] == "bc3"
    assert np.array_equal(item["abd"], g["item3_abd"]) and np.array_equal(item["tnf"], g["item3_tnf"]) and len(d) == 37
    # device tensors given directly (what the feature kernels hand over) behave the same
    d2 = Data(d.bc, torch.from_numpy(g["abd_in"]).to(device), torch.from_numpy(g["tnf_in"]).to(device))
    assert np.array_equal(d2.abd, g["abd"]) and np.array_equal(d2.weights, g["weights"])


@pytest.mark.gpu
def test_fused_row_normalisation_is_bit_identical():
    """int32 device matrices (what the feature kernels leave) take ONE kernel (pg_normalize_rows) instead of the float64 torch
    passes: same bits as the reference's vectors, as the oracle and as the torch path -- zero rows, negative entries, ragged widths"""
    g = np.load(os.path.join(GOLDEN, "data_g4.npz"))
    a32, t32 = torch.from_numpy(g["abd_in"].astype(np.int32)).cuda(), torch.from_numpy(g["tnf_in"].astype(np.int32)).cuda()
    assert np.array_equal(a32.cpu().numpy(), g["abd_in"]) and np.array_equal(t32.cpu().numpy(), g["tnf_in"])     # (the golden inputs are counts)
    d = Data(np.arange(37), a32, t32)
    assert d.abd_dev.dtype == torch.float32 and d.weights.dtype == np.float64
    assert np.array_equal(d.abd, g["abd"]) and np.array_equal(d.tnf, g["tnf"]) and np.array_equal(d.weights, g["weights"])
    rs = np.random.RandomState(4)
    for n, va, vt in ((1, 1, 1), (5, 400, 136), (1000, 63, 65), (257, 512, 32)):
        abd = rs.randint(0, 2000, size=(n, va)).astype(np.int32)
        tnf = rs.randint(-50, 3000, size=(n, vt)).astype(np.int32)
        abd[rs.rand(n) < 0.2] = 0                                    # rows without any k-mer
        abd[rs.rand(n) < 0.1, 0] = 2_000_000_000                     # sums beyond 32 bits
        fused = Data(np.arange(n), torch.from_numpy(abd).cuda(), torch.from_numpy(tnf).cuda())
        plain = Data(np.arange(n), torch.from_numpy(abd.astype(np.int64)).cuda(), torch.from_numpy(tnf.astype(np.int64)).cuda())
        oa, ot, ow = oracle.data_normalize(abd, tnf)
        for x, y, z in ((fused.abd, plain.abd, oa), (fused.tnf, plain.tnf, ot), (fused.weights, plain.weights, ow)):
            assert np.array_equal(x, y) and np.array_equal(x, z)


@pytest.mark.parametrize("fixture,hidden,n_classes", FIXTURES)
@pytest.mark.parametrize("device", DEVICES)
def test_network_matches_reference_vectors(device, fixture, hidden, n_classes):
    g, state = _g5(fixture)
    net = (VaritionalAutoEncoder(400, 136, hidden_sizes=hidden) if hidden else VaritionalAutoEncoder(400, 136)).to(device)
    if hidden is None:
        assert tuple(state["encoder.0.weight"].shape) == (512, 536) and tuple(state["encoder.4.weight"].shape) == (512, 512)
    assert set(net.state_dict()) == set(state)                       # the reference's state_dict keys
    net.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    net.eval()
    abd, tnf, eps = (torch.from_numpy(g[k]).to(device) for k in ("abd", "tnf", "epsilon"))
    scale = np.abs(g["mu"]).max()
    with torch.no_grad():
        mu = net.emebdding(abd, tnf).cpu().numpy()
        out = net(abd, tnf, eps)
    assert np.abs(mu - g["mu"]).max() <= 1e-5 * scale
    assert np.abs(mu - oracle.vae_embedding(state, g["abd"], g["tnf"])).max() <= 1e-5 * scale
    for key, ref in (("mu", "fwd_mu"), ("logsigma", "fwd_logsigma"), ("abd_rec", "fwd_abd_rec"), ("tnf_rec", "fwd_tnf_rec")):
        assert np.abs(out[key].cpu().numpy() - g[ref]).max() <= 1e-5 * max(1e-3, np.abs(g[ref]).max()), key
    vn = VAENET(400, 136, 32, n_classes, 1, device != "cpu", 1, 0.005, 0.2, 0.1, 0.015, 0.0001)
    assert np.isclose(vn.wa, float(g["wa"])) and np.isclose(vn.wt, float(g["wt"])) and np.isclose(vn.w_kl, float(g["w_kl"]))
    losses = vn.unlabeled_loss(out)
    for key, ref in (("total", "loss_total"), ("abd_rec", "loss_abd"), ("tnf_rec", "loss_tnf"), ("kl_loss", "loss_kl")):
        assert abs(losses[key].item() - float(g[ref])) <= 1e-5 * abs(float(g[ref])), key
    o = oracle.vae_forward_loss(state, g["abd"], g["tnf"], g["epsilon"], vn.wa, vn.wt, vn.w_kl)
    assert abs(o["total"] - float(g["loss_total"])) <= 1e-5 * abs(float(g["loss_total"]))


@pytest.mark.parametrize("device", DEVICES)
def test_train_writes_the_reference_files_and_resumes(device, tmp_path):
    rs = np.random.RandomState(0)
    n = 300
    abd = rs.poisson(2.0, (n, 400)); tnf = rs.poisson(30.0, (n, 136))
    names = np.array([f"b{i}" for i in range(n)], dtype=object)
    data = Data(names, abd, tnf, device=device)
    np.random.seed(2021); torch.manual_seed(2021)
    train = weighted_batches(data, 64)
    test = weighted_batches(data, 64, num_samples=int(n * 0.7), replacement=False)
    original = shuffled_batches(data, 64)
    assert len(train) == 5 and len(test) == 4 and len(original) == 5
    idx = np.concatenate([b["bc"] for b in original])
    assert sorted(idx) == sorted(names)                              # a permutation
    vae = VAENET(400, 136, 32, 5, 3, device != "cpu", 1, 0.005, 0.2, 0.1, 0.015, 0.0001)
    model = tmp_path / "2.vae"
    with pytest.raises(Exception, match="model path not exist"):
        vae.train(train, test, original, str(model), 20)
    model.mkdir()
    vae.train(train, test, original, str(model), 20)
    for f in ("train_model.pk", "latent.npz", "barcodes.npz", "model_finished"):
        assert (model / f).is_file(), f
    latent = np.load(model / "latent.npz")["arr_0"]; bcs = np.load(model / "barcodes.npz")["arr_0"]
    assert latent.shape == (n, 32) and latent.dtype == np.float32 and list(bcs) == list(names)
    state = {k: v.cpu().numpy() for k, v in torch.load(model / "train_model.pk", map_location="cpu").items()}
    want = oracle.vae_embedding(state, data.abd, data.tnf)
    assert np.abs(latent - want).max() <= 1e-5 * np.abs(want).max()
    # resume: existing checkpoint and latents are kept
    before = (model / "latent.npz").stat().st_mtime_ns
    VAENET(400, 136, 32, 5, 3, device != "cpu", 1, 0.005, 0.2, 0.1, 0.015, 0.0001).train(train, test, original, str(model), 20)
    assert (model / "latent.npz").stat().st_mtime_ns == before


def test_sampler_draws_are_the_reference_draws():
    """CustomWeightedRandomSampler semantics (utils.py:13-21): np.random.choice over p = w / sum(w) from numpy's global
    generator, with and without replacement"""
    w = np.array([0.1, 0.4, 0.2, 0.05, 0.25])

    class D:
        weights, abd_dev, tnf_dev, bc = w, torch.zeros(5, 2), torch.zeros(5, 2), np.arange(5)
        def __len__(self): return 5
    np.random.seed(7)
    got = np.concatenate([b["bc"] for b in weighted_batches(D(), 2)])
    np.random.seed(7)
    assert list(got) == list(np.random.choice(range(5), size=5, p=w / w.sum(), replace=True))
    np.random.seed(8)
    got = np.concatenate([b["bc"] for b in weighted_batches(D(), 2, num_samples=3, replacement=False)])
    np.random.seed(8)
    assert list(got) == list(np.random.choice(range(5), size=3, p=w / w.sum(), replace=False))


@pytest.mark.gpu
def test_graph_captured_training_learns_like_the_eager_loop(tmp_path, monkeypatch):
    """the hipGraph replay of the training step and of the validation forward is the same computation as the eager loop:
    same files, same checkpoint keys, a validation loss in the same place (dropout masks and epsilon differ by design)"""
    rs = np.random.RandomState(1)
    n, k = 4096, 6
    proto_a, proto_t = rs.rand(k, 400) ** 4, rs.rand(k, 136) + 0.2
    which = rs.randint(0, k, n)
    abd, tnf = rs.poisson(proto_a[which] * 300), rs.poisson(proto_t[which] * 400)
    names = np.array([f"b{i}" for i in range(n)], dtype=object)
    finals = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PG_TRAIN_GRAPH", mode)
        data = Data(names, abd, tnf, device="cuda:0")
        np.random.seed(2021); torch.manual_seed(2021)
        loaders = (weighted_batches(data, 256), weighted_batches(data, 256, num_samples=int(n * 0.7), replacement=False), shuffled_batches(data, 256))
        vae = VAENET(400, 136, 32, k, 6, True, 1, 0.005, 0.2, 0.1, 0.015, 0.0001)
        model = tmp_path / f"vae{mode}"
        model.mkdir()
        vae.train(*loaders, str(model), 50)
        assert (vae._val_step is not None) == (mode == "1")
        np.random.seed(7)
        finals[mode] = vae._validate(weighted_batches(data, 256, num_samples=2048, replacement=False))
        latent = np.load(model / "latent.npz")["arr_0"]
        assert latent.shape == (n, 32) and np.isfinite(latent).all()
        state = {k2: v.cpu().numpy() for k2, v in torch.load(model / "train_model.pk", map_location="cpu").items()}
        want = oracle.vae_embedding(state, data.abd, data.tnf)
        assert np.abs(latent - want).max() <= 1e-5 * np.abs(want).max()
    assert abs(finals["1"] - finals["0"]) <= 0.1 * abs(finals["0"]), finals


@pytest.mark.gpu
def test_device_sampling_has_the_reference_distribution_and_numpy_mode_its_sequence(monkeypatch):
    """device mode: weighted draws by torch.multinomial on the GPU (utils.py:13-21's distribution, another sequence), seeded from
    numpy's generator; numpy mode: the reference's very sequence"""
    from pangaea_amd.loader import weighted_batches
    rs = np.random.RandomState(1)
    n = 4000
    w = rs.rand(n) ** 3 + 1e-3

    class D:
        weights, abd_dev, tnf_dev, bc = w, torch.zeros(n, 2, device="cuda:0"), torch.zeros(n, 2, device="cuda:0"), np.arange(n)
        def __len__(self): return n
    # with replacement: frequencies follow the weights
    np.random.seed(3)
    draws = np.concatenate([np.asarray(b["bc"]) for _ in range(50) for b in weighted_batches(D(), 512, mode="device")])
    freq = np.bincount(draws, minlength=n) / len(draws)
    p = w / w.sum()
    top = np.argsort(p)[-200:]
    assert abs(freq[top].sum() - p[top].sum()) < 0.01 and np.corrcoef(freq, p)[0, 1] > 0.97
    # without replacement: distinct rows, heavy rows come early (successive sampling), reproducible under the numpy seed
    np.random.seed(4)
    a = np.concatenate([np.asarray(b["bc"]) for b in weighted_batches(D(), 512, num_samples=2800, replacement=False, mode="device")])
    np.random.seed(4)
    b = np.concatenate([np.asarray(b["bc"]) for b in weighted_batches(D(), 512, num_samples=2800, replacement=False, mode="device")])
    assert len(a) == 2800 and len(set(a.tolist())) == 2800 and np.array_equal(a, b)
    assert p[a[:400]].mean() > 2 * p[a[-400:]].mean()
    heavy = np.argsort(p)[-100:]
    assert np.isin(heavy, a).all()
    # numpy mode keeps the reference's sequences (also selectable by the environment)
    monkeypatch.setenv("PANGAEA_SAMPLING", "numpy")
    np.random.seed(8)
    got = np.concatenate([np.asarray(b["bc"]) for b in weighted_batches(D(), 512, num_samples=300, replacement=False)])
    np.random.seed(8)
    assert list(got) == list(np.random.choice(range(n), size=300, p=w / w.sum(), replace=False))


@pytest.mark.parametrize("device", DEVICES)
def test_blas_warm_up_is_a_no_op_without_a_gpu_and_a_thread_with_one(device, monkeypatch):
    """runtime.warm_blas: the GEMM library's one-time initialisation on a helper thread (nothing to do on the CPU; switched off by
    PANGAEA_WARM_BLAS=0); an encode that runs beside it or behind it gives the same latents"""
    from pangaea_amd import runtime
    assert runtime.warm_blas("cpu") is None
    monkeypatch.setenv("PANGAEA_WARM_BLAS", "0")
    assert runtime.warm_blas(device) is None
    monkeypatch.delenv("PANGAEA_WARM_BLAS")
    t = runtime.warm_blas(device)
    if torch.device(device).type != "cuda":
        assert t is None
        return
    x = torch.ones((64, 536), device=device)
    w = torch.full((512, 536), 0.5, device=device)
    y = torch.nn.functional.linear(x, w)                 # (beside the helper thread)
    t.join()
    assert not t.is_alive() and torch.equal(y, torch.nn.functional.linear(x, w)) and float(y[0, 0]) == 268.0
