"""Device-side weight packing (phnn_update_weights_dev, SURVEY 8 row f4 / VERDICT round 2 item 6): the image packed by
the one-workgroup kernel from a GPU-resident blob equals the host-packed image -- bit for bit for pHNN and ODEFunc
models; for canonical models everywhere except the eleven constants that go through exp / log1p (softplus(R_diag_raw),
sigmoid, the mass-matrix constants), which may differ in the last bit between the two math libraries."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu

MODELS = sorted(ol.MODELS) + [ol.TRAINED] + sorted(ol.ACT_MODELS)


@pytest.fixture(scope="module")
def torch():
    import torch as t
    if not t.cuda.is_available():
        pytest.skip("needs a GPU")
    return t


def perturbed(w, rng, scale):
    return {k: (v * (1.0 + scale * rng.standard_normal(v.shape))).astype(np.float32) if v.dtype.kind == "f" else v
            for k, v in w.items()}


@pytest.mark.parametrize("name", MODELS)
@pytest.mark.parametrize("mode", ["default", "f32", "bf16x3"])
def test_device_packed_image_equals_host_packed(torch, name, mode):
    from phnn_mpc_amd import weights
    from phnn_mpc_amd.engine import PhnnError, RolloutEngine
    w = ol.load_weights(name)
    act = ol.ACT_MODELS.get(name, "tanh")
    try:
        eng = RolloutEngine(w, matmul=mode, activation=act)
    except PhnnError:
        pytest.skip(f"{name}: no {mode} kernels")
    rng = np.random.default_rng(5)
    for scale in (0.0, 0.05, 3.0):  # the same weights, nearby ones, and ones that move the power-of-two scales
        w2 = perturbed(w, rng, scale)
        _, blob = weights.pack_state_dict(w2, kind=eng.kind, activation=act)
        eng.update_weights(w2)
        host = eng.read_image()
        eng.update_weights(w)  # back to other values, so that the device pack has something to overwrite
        eng.update_weights_dev(torch.tensor(blob, device=eng.device))
        dev = eng.read_image()
        diff = np.flatnonzero(host.view(np.uint32) != dev.view(np.uint32))
        if eng.kind == 1:  # canonical: <= 11 libm-dependent constants, within one unit in the last place
            assert diff.size <= 11, (name, mode, scale, diff.size)
            assert np.allclose(host[diff], dev[diff], rtol=2.5e-7, atol=0.0)
        else:
            assert diff.size == 0, (name, mode, scale, diff[:8], host[diff[:8]], dev[diff[:8]])


def test_training_step_on_gpu_parameters_uses_the_device_packer(torch, monkeypatch):
    """A module whose parameters live on the GPU re-packs through update_weights_dev after an optimizer step (no host
    packing), and computes what a module re-packed on the host computes."""
    import os
    from phnn_mpc_amd.engine import RolloutEngine
    from phnn_mpc_amd.models import pHNN
    cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "cartpole_mpc.yaml")
    w = ol.load_weights("phnn_cartpole")
    dev = torch.device("cuda:0")
    calls = {"dev": 0, "host": 0}
    orig_dev, orig_host = RolloutEngine.update_weights_dev, RolloutEngine.update_weights
    monkeypatch.setattr(RolloutEngine, "update_weights_dev", lambda self, b: (calls.__setitem__("dev", calls["dev"] + 1), orig_dev(self, b))[1])
    monkeypatch.setattr(RolloutEngine, "update_weights", lambda self, sd: (calls.__setitem__("host", calls["host"] + 1), orig_host(self, sd))[1])
    outs = []
    for where in ("cuda", "cpu"):
        m = pHNN(cfg)
        m.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
        if where == "cuda":
            m = m.to(dev)
        m.use_device(dev)
        opt = torch.optim.SGD(m.parameters(), lr=1e-3)
        x = torch.tensor([[0.1, 0.2, -0.3, 0.4]], device=dev, requires_grad=True)
        u = torch.tensor([[0.5]], device=dev)
        for _ in range(3):
            opt.zero_grad()
            dx, H = m(x, u)
            (dx.square().sum() + H.sum()).backward()
            opt.step()
        dx, H = m(x, u)
        outs.append((dx.detach().cpu().numpy(), H.detach().cpu().numpy()))
        if where == "cuda":
            assert calls["dev"] == 3 and calls["host"] == 0, calls
    assert calls["host"] == 3
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


NAMED = ([("golden_m2.npz", n) for n in ol.M2_MODELS] + [("golden_m34.npz", n) for n in ("phnn_m3_fix", "phnn_m4_gnet", "canonical_m3")] +
         [("golden_mass.npz", n) for n in ol.MASS_TYPES])


@pytest.mark.parametrize("fname,name", NAMED)
def test_device_packed_image_equals_host_packed_multi_input_and_mass_models(torch, fname, name):
    """The same for the models with several control inputs (G is (n, m)) and the MassMatrixNetwork variants (mass constants /
    M_net.mlp in the image)."""
    from phnn_mpc_amd import weights
    from phnn_mpc_amd.engine import RolloutEngine
    _, ws = ol.load_named_golden(fname)
    w = ws[name]
    eng = RolloutEngine(w)
    rng = np.random.default_rng(11)
    for scale in (0.0, 0.05):
        w2 = perturbed(w, rng, scale)
        _, blob = weights.pack_state_dict(w2, kind=eng.kind)
        eng.update_weights(w2)
        host = eng.read_image()
        eng.update_weights(w)
        eng.update_weights_dev(torch.tensor(blob, device=eng.device))
        dev = eng.read_image()
        diff = np.flatnonzero(host.view(np.uint32) != dev.view(np.uint32))
        if eng.kind == 1:  # canonical: the libm-dependent constants (softplus / sigmoid of R_diag_raw, mass constants)
            assert diff.size <= 14, (name, scale, diff.size)
            assert np.allclose(host[diff], dev[diff], rtol=2.5e-7, atol=0.0)
        else:
            assert diff.size == 0, (name, scale, diff[:8])
