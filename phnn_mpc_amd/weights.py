"""state_dict -> (phnn_desc, float32 blob) packing, layout documented in include/phnn_mpc.h.

The keys read here are exactly the reference's state_dict keys:
  pHNN            J, G_fixed | G_net.net.*, R_net.net.{0,2,..}.{weight,bias}, H_net.net.*   (src/pHNN.py:22-38)
  pHNN_Canonical  R_diag_raw, J (ignored: fixed canonical), G, M_net.{log_a,b,log_c}, H_net.net.*
                  (src/pHNN_canonical.py:57-110)
  ODEFunc         network.{0,2,..}.{weight,bias}                                         (src/baseline_node.py:60-75)
Checkpoints may be a bare state_dict or {'model_state_dict': ...} (scripts/run_cartpole_mpc.py:41-44).
"""
import numpy as np

from . import _capi


def _np(v):
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(v, dtype=np.float32))


def unwrap_checkpoint(obj):
    if isinstance(obj, dict) and "model_state_dict" in obj:
        return obj["model_state_dict"]
    return obj


def _mlp_layers(sd, prefix):
    """Linear layers of an nn.Sequential stored under `prefix` (indices 0,2,4,.. with Tanh between)."""
    idx = sorted({int(k[len(prefix):].split(".")[0]) for k in sd if k.startswith(prefix) and k.endswith(".weight")})
    if not idx:
        raise KeyError(f"no Linear layers under '{prefix}' in state_dict")
    layers = []
    for i in idx:
        W, b = _np(sd[f"{prefix}{i}.weight"]), _np(sd[f"{prefix}{i}.bias"])
        if W.ndim != 2 or b.shape != (W.shape[0],):
            raise ValueError(f"{prefix}{i}: unexpected shapes {W.shape} / {b.shape}")
        layers.append((W, b))
    for (W0, _), (W1, _) in zip(layers[:-1], layers[1:]):
        if W1.shape[1] != W0.shape[0]:
            raise ValueError(f"{prefix}: layer widths do not chain ({W0.shape} -> {W1.shape}); LayerNorm/other "
                             "layers are not supported")
    return layers


def _mass_kind(sd):
    """Which mass matrix a pHNN_Canonical state_dict carries (src/pHNN_canonical.py:67-86) -> (mass_type, mlp layers)."""
    if "M_net.log_a" in sd:
        for k in ("M_net.b", "M_net.log_c"):
            if k not in sd:
                raise ValueError(f"pHNN_Canonical state_dict: CartPoleMassMatrix parameter {k} is missing")
        return _capi.MASS_CARTPOLE, None
    if "M_net.L_tril" in sd:
        if _np(sd["M_net.L_tril"]).shape != (2, 2):
            raise ValueError("MassMatrixNetwork: only q_dim = 2 is supported")
        return _capi.MASS_CONSTANT, None
    if any(k.startswith("M_net.mlp.") for k in sd):
        Ml = _mlp_layers(sd, "M_net.mlp.")
        out, qd = Ml[-1][0].shape[0], Ml[0][0].shape[1]
        if qd != 2 or out not in (2, 3):
            raise ValueError("MassMatrixNetwork: only q_dim = 2 (2 diagonal or 3 Cholesky outputs) is supported")
        return (_capi.MASS_DIAGONAL if out == 2 else _capi.MASS_FULL), Ml
    raise ValueError("pHNN_Canonical state_dict without mass-matrix parameters (M_net.log_a/b/log_c, M_net.L_tril "
                     "or M_net.mlp.*)")


def _flat(layers):
    out = []
    for W, b in layers:
        out += [W.ravel(), b.ravel()]
    return out


def detect_kind(sd):
    keys = set(sd.keys())
    if any(k.startswith("network.") for k in keys):
        return _capi.MODEL_ODEFUNC
    if "R_diag_raw" in keys:
        return _capi.MODEL_CANONICAL
    if "J" in keys and any(k.startswith("R_net.") for k in keys):
        return _capi.MODEL_PHNN
    raise ValueError("state_dict is neither a pHNN, a pHNN_Canonical nor an ODEFunc")


def _activation_code(activation):
    """'tanh' / 'Tanh' / 'nn.Tanh' / torch.nn.Tanh -> PHNN_ACT_TANH, likewise SiLU and ReLU; anything else ->
    PHNN_ACT_OTHER (refused by phnn_create).  The state_dict itself does not say which activation a checkpoint was
    trained with."""
    name = activation if isinstance(activation, str) else getattr(activation, "__name__", str(activation))
    return _capi.ACTIVATIONS.get(name.split(".")[-1].lower(), _capi.ACT_OTHER)


def pack_state_dict(sd, kind=None, state_dim=None, input_dim=None, activation="tanh"):
    """Return (Desc, blob) for a reference state_dict (numpy arrays or torch tensors).

    activation: the activation of the MLPs the weights were trained with (the reference resolves it from the YAML,
    src/pHNN.py:41).  Only Tanh has kernels; any other value makes phnn_create fail instead of silently running a
    SiLU/ReLU checkpoint -- same keys, same shapes -- as a Tanh network."""
    sd = unwrap_checkpoint(sd)
    if kind is None:
        kind = detect_kind(sd)
    d = _capi.Desc()
    d.kind = kind
    d.activation = _activation_code(activation)
    parts = []
    if kind == _capi.MODEL_PHNN:
        J = _np(sd["J"])
        n = J.shape[0]
        H = _mlp_layers(sd, "H_net.net.")
        Rl = _mlp_layers(sd, "R_net.net.")
        if H[0][0].shape[1] != n or H[-1][0].shape[0] != 1 or Rl[-1][0].shape[0] != n * n:
            raise ValueError("pHNN state_dict: H_net/R_net shapes inconsistent with J")
        fixed = "G_fixed" in sd
        if fixed:
            G = _np(sd["G_fixed"])
            m = G.shape[1]
            Gl = []
        else:
            Gl = _mlp_layers(sd, "G_net.net.")
            m = Gl[-1][0].shape[0] // n
        d.n, d.m, d.fixed_G = n, m, int(fixed)
        d.h_net = _capi.MlpShape.of([W.shape[0] for W, _ in H[:-1]])
        d.r_net = _capi.MlpShape.of([W.shape[0] for W, _ in Rl[:-1]])
        parts.append(J.ravel())
        if fixed:
            parts.append(G.ravel())
        parts += _flat(Rl) + _flat(H)
        if not fixed:
            d.g_net = _capi.MlpShape.of([W.shape[0] for W, _ in Gl[:-1]])
            parts += _flat(Gl)
    elif kind == _capi.MODEL_CANONICAL:
        Rd = _np(sd["R_diag_raw"])
        n = Rd.shape[0]
        G = _np(sd["G"])
        m = G.shape[1]
        H = _mlp_layers(sd, "H_net.net.")
        d.n, d.m, d.fixed_G = n, m, 1
        d.h_net = _capi.MlpShape.of([W.shape[0] for W, _ in H[:-1]])
        parts += [Rd.ravel(), G.ravel()]
        mass, Ml = _mass_kind(sd)
        d.mass_type = mass
        if mass == _capi.MASS_CARTPOLE:  # CartPoleMassMatrix (src/mass_matrix.py:263-268)
            parts.append(np.array([_np(sd["M_net.log_a"]).item(), _np(sd["M_net.b"]).item(),
                                   _np(sd["M_net.log_c"]).item()], np.float32))
        elif mass == _capi.MASS_CONSTANT:  # MassMatrixNetwork 'constant': L_tril (q_dim, q_dim)
            parts.append(_np(sd["M_net.L_tril"]).ravel())
        else:  # 'diagonal' / 'full': M_net.mlp
            d.m_net = _capi.MlpShape.of([W.shape[0] for W, _ in Ml[:-1]])
            parts += _flat(Ml)
        parts += _flat(H)
    elif kind == _capi.MODEL_ODEFUNC:
        L = _mlp_layers(sd, "network.")
        n = L[-1][0].shape[0] if state_dim is None else int(state_dim)
        m = L[0][0].shape[1] - n if input_dim is None else int(input_dim)
        if L[0][0].shape[1] != n + m or L[-1][0].shape[0] != n:
            raise ValueError("ODEFunc state_dict: first/last layer do not match (state_dim, action_dim)")
        d.n, d.m, d.fixed_G = n, m, 1
        d.h_net = _capi.MlpShape.of([W.shape[0] for W, _ in L[:-1]])
        parts += _flat(L)
    else:
        raise ValueError(f"unknown model kind {kind}")
    blob = np.ascontiguousarray(np.concatenate(parts).astype(np.float32))
    return d, blob


def blob_layout(sd, kind=None):
    """[(state_dict key, offset, shape)] of every tensor inside the float32 weight blob of pack_state_dict, in blob
    order.  The gradient blob the weight-gradient kernels return has the same layout; entries that are buffers of
    the reference modules (G_fixed; the canonical model's G) or constants to autograd (M_net.*) stay zero there."""
    sd = unwrap_checkpoint(sd)
    if kind is None:
        kind = detect_kind(sd)
    out, off = [], 0

    def add(key, shape):
        nonlocal off
        cnt = 1
        for d in shape:
            cnt *= int(d)
        out.append((key, off, tuple(int(d) for d in shape)))
        off += cnt

    def add_mlp(prefix):
        idx = sorted({int(k[len(prefix):].split(".")[0]) for k in sd if k.startswith(prefix) and k.endswith(".weight")})
        for i in idx:
            add(f"{prefix}{i}.weight", _np(sd[f"{prefix}{i}.weight"]).shape)
            add(f"{prefix}{i}.bias", _np(sd[f"{prefix}{i}.bias"]).shape)

    if kind == _capi.MODEL_PHNN:
        add("J", _np(sd["J"]).shape)
        if "G_fixed" in sd:
            add("G_fixed", _np(sd["G_fixed"]).shape)
        add_mlp("R_net.net.")
        add_mlp("H_net.net.")
        if "G_fixed" not in sd:
            add_mlp("G_net.net.")
    elif kind == _capi.MODEL_CANONICAL:
        add("R_diag_raw", _np(sd["R_diag_raw"]).shape)
        add("G", _np(sd["G"]).shape)
        mass, _ = _mass_kind(sd)
        if mass == _capi.MASS_CARTPOLE:
            for k in ("M_net.log_a", "M_net.b", "M_net.log_c"):
                add(k, ())
        elif mass == _capi.MASS_CONSTANT:
            add("M_net.L_tril", (2, 2))
        else:
            add_mlp("M_net.mlp.")
        add_mlp("H_net.net.")
    elif kind == _capi.MODEL_ODEFUNC:
        add_mlp("network.")
    else:
        raise ValueError(f"unknown model kind {kind}")
    return out


def unpack_grad_blob(sd, blob, kind=None, layout=None):
    """gradient blob (P,) -> {state_dict key: array of the parameter's shape} (numpy or torch, views of `blob`)."""
    res = {}
    for key, off, shape in (layout if layout is not None else blob_layout(sd, kind)):
        cnt = 1
        for d in shape:
            cnt *= d
        res[key] = blob[off:off + cnt].reshape(shape)
    return res
