"""Drop-in model classes: same constructor arguments, state_dict keys and call signatures as the reference's
nn.Modules, with forward() served by the gfx950 kernels through RolloutEngine.

  pHNN            <- src/pHNN.py:12-100            model(x,u) -> (dx, H)
  pHNN_Canonical  <- src/pHNN_canonical.py:40-290  model(y,u) -> (dy, H, None)
  ODEFunc         <- src/baseline_node.py:19-116   f(t, state) with the current_action attribute

The modules only hold parameters (so checkpoints of the reference load with load_state_dict) and hand a
packed copy of them to the engine; no dynamics arithmetic is done in torch.  Unlike the reference, the
input does not need requires_grad and the call works under torch.no_grad() (SURVEY.md quirk 6 is an
artefact of autograd.grad inside forward).  First-order gradients w.r.t. x and u flow through the returned
dx (custom autograd.Function backed by phnn_model_vjp); H is returned without a graph.
"""
import math

import torch
import torch.nn as nn
import yaml

from . import _capi

DEFAULT_DEVICE = "cuda:0"


def _activation(name):
    return getattr(nn, name.split(".")[-1])  # same resolution rule as src/pHNN.py:41


class MLP(nn.Module):
    """Parameter container with the reference's layout: self.net = Sequential(Linear, act, ..., Linear)
    (src/NN.py:6-40), so state_dict keys read '<name>.net.<2k>.weight'."""

    def __init__(self, input_dim, output_dim, hidden_sizes=(128, 128), activation=nn.SiLU, dropout=0.0,
                 layer_norm=False, bias=True):
        super().__init__()
        self.activation_name = activation.__name__
        self.plain = (not layer_norm) and dropout == 0 and bias
        mods, last = [], input_dim
        for h in hidden_sizes:
            mods.append(nn.Linear(last, h, bias=bias))
            if layer_norm:
                mods.append(nn.LayerNorm(h))
            mods.append(activation())
            if dropout > 0:
                mods.append(nn.Dropout(dropout))
            last = h
        mods.append(nn.Linear(last, output_dim, bias=bias))
        self.net = nn.Sequential(*mods)
        for m in self.net:  # Kaiming-uniform(a=sqrt 5) weights, U(+-1/sqrt(fan_in)) biases
            if isinstance(m, nn.Linear):
                nn.init.kaiming_uniform_(m.weight, a=math.sqrt(5))
                if m.bias is not None:
                    bound = 1 / math.sqrt(m.in_features) if m.in_features > 0 else 0.0
                    nn.init.uniform_(m.bias, -bound, bound)


def _mlp_from(params, input_dim, output_dim):
    return MLP(input_dim, output_dim, hidden_sizes=tuple(params["hidden_sizes"]), activation=_activation(params["activation"]),
               dropout=params["dropout"], layer_norm=params["layer_norm"], bias=params["bias"])


def packed_engine(owner):
    """The engine as the LAST `owner.engine` access left it -- no parameter-change check (one `owner.engine` per public
    operation does that; walking the module tree for the fingerprint costs tens of microseconds each time)."""
    eng = getattr(owner, "_engine", None)
    return eng if eng is not None else owner.engine


def live_engine(eng):
    """The engine a forward pass ran on, for its backward pass."""
    if getattr(eng, "h", True) is None:
        raise RuntimeError("the model's engine was rebuilt (refresh_engine / load_state_dict / use_device) between "
                           "forward and backward")
    return eng


def grad_params(owner):
    """Parameters of an engine-backed module that (a) live in the weight blob and (b) require grad, in blob order:
    ([state_dict keys], [nn.Parameter]).  Buffers (G_fixed, the canonical J / G) are not parameters."""
    named = dict(owner.named_parameters())
    keys = [k for k, _, _ in packed_engine(owner).layout if k in named and named[k].requires_grad]
    return keys, [named[k] for k in keys]


def split_param_grads(engine, grad_theta, keys, device, owner=None, points=None):
    """gradient blob -> tuple of per-parameter gradients in the order of `keys` (on `device`).  owner + points
    (B, H, integrator of the wgrad call just made): a canonical model with a MassMatrixNetwork gets the mass network's
    parameter gradients from one autograd pass of that module over the recorded evaluation points (the kernels leave
    them at zero and record q and the cotangent of M(q) instead; include/phnn_mpc.h: phnn_wgrad_record_info)."""
    named = dict(engine.named_grads(grad_theta))
    mass = getattr(owner, "M_net", None) if owner is not None else None
    if (isinstance(mass, MassMatrixNetwork) and points is not None and hasattr(engine, "mass_cotangents")
            and any(k.startswith("M_net.") for k in keys)):
        q, Mb = engine.mass_cotangents(*points)
        for k, g in mass_param_grads(mass, q, Mb).items():
            named["M_net." + k] = g
    return tuple(named[k].to(device) for k in keys)


def mass_param_grads(mass, q, Mbar):
    """d sum_p <Mbar_p, M(q_p)> / d theta_M for a MassMatrixNetwork: {parameter name: gradient} on q's device (float32)."""
    params = {k: p.detach().to(q.device, torch.float32).requires_grad_(True) for k, p in mass.named_parameters()}
    with torch.enable_grad():
        M = torch.func.functional_call(mass, params, (q.detach(),))
        grads = torch.autograd.grad(M, list(params.values()), grad_outputs=Mbar.detach(), allow_unused=True)
    return {k: (torch.zeros_like(p) if g is None else g) for (k, p), g in zip(params.items(), grads)}


_warned_no_wgrad = set()


def _no_wgrad_warning(owner):
    name = type(owner).__name__
    if name not in _warned_no_wgrad:
        _warned_no_wgrad.add(name)
        import warnings
        warnings.warn(f"{name}: the engine has no weight-gradient kernels for this model family; backward() leaves "
                      "its parameters without .grad.  Gradients w.r.t. x and u of the dx output are exact; a cotangent on "
                      "the H output raises NotImplementedError (the plain VJP kernel has no H input)", RuntimeWarning, stacklevel=3)


class _ModelFn(torch.autograd.Function):
    """model(x,u) -> (dx, H) on the engine.  backward: VJP w.r.t. x and u (phnn_model_vjp) and, when parameters
    require grad, the parameter gradients of the same cotangents (phnn_model_wgrad) -- what autograd does for one
    model call of the reference (src/pHNN.py:52-100)."""

    @staticmethod
    def forward(ctx, x, u, owner, keys, *params):
        eng = ctx.eng = packed_engine(owner)  # call_model_fn has just checked it against the parameters
        xd, ud = x.detach().to(eng.device, torch.float32), u.detach().to(eng.device, torch.float32)
        dx, H = eng.forward(xd, ud)
        ctx.owner, ctx.dev, ctx.keys = owner, x.device, keys
        ctx.pdev = params[0].device if params else None
        ctx.save_for_backward(xd, ud)
        ctx.set_materialize_grads(False)  # an unused output arrives as None, not as a tensor of zeros
        return dx.to(x.device), H.to(x.device)

    @staticmethod
    def backward(ctx, gdx, gH):
        xd, ud = ctx.saved_tensors
        eng = live_engine(ctx.eng)
        gdx = torch.zeros_like(xd) if gdx is None else gdx.to(eng.device, torch.float32)
        want_params = bool(ctx.keys) and any(ctx.needs_input_grad[4:])
        if want_params or (gH is not None and eng.has_wgrad):
            gth, xb, ub = eng.model_wgrad(xd, ud, gdx, None if gH is None else gH.to(eng.device, torch.float32))
            pg = (split_param_grads(eng, gth, ctx.keys, ctx.pdev, ctx.owner, (xd.shape[0], 0, "euler")) if want_params
                  else (None,) * len(ctx.keys))
            return (xb.to(ctx.dev), ub.to(ctx.dev), None, None) + pg
        if gH is not None and getattr(eng, "kind", None) != _capi.MODEL_ODEFUNC and bool((gH != 0).any()):
            # only the record-emitting VJP (has_wgrad) takes a cotangent on H; dropping it would return wrong x / u
            # gradients for a loss through H (e.g. an energy-anchor term) without any sign of it
            raise NotImplementedError("backward through the H output needs the weight-gradient kernels, which this model "
                                      "variant / matmul mode does not have (has_wgrad is False)")
        xb, ub = eng.vjp(xd, ud, gdx)
        return (xb.to(ctx.dev), ub.to(ctx.dev), None, None) + (None,) * len(ctx.keys)


def call_model_fn(owner, x, u):
    """_ModelFn with the module's trainable parameters as explicit inputs (so backward can hand them gradients)."""
    keys, params = [], []
    eng = owner.engine  # re-packs the weights if a parameter has changed
    if torch.is_grad_enabled():
        if eng.has_wgrad:
            keys, params = grad_params(owner)
        elif any(p.requires_grad for p in owner.parameters()):
            _no_wgrad_warning(owner)
    return _ModelFn.apply(x, u, owner, keys, *params)


class _EngineBacked(nn.Module):
    """Shared plumbing: lazily built engine, invalidated when parameters are (re)loaded."""

    def __init__(self):
        super().__init__()
        self._engine = None
        self._engine_device = DEFAULT_DEVICE

    def _fingerprint(self):
        """Cheap identity of the current parameter values: (storage pointer, in-place version counter) of every
        parameter and buffer.  An optimizer step, copy_(), an in-place edit of the parameter itself or a re-assigned
        .data changes it.  NOT caught: in-place edits made THROUGH `p.data` (p.data.mul_(..), p.data -= ..) -- they bump
        neither the pointer nor p._version; call refresh_engine() after such edits."""
        return tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))

    @property
    def engine(self):
        """The engine holding the packed weights.  Built lazily; when a parameter has changed since the weights were
        packed (optimizer step, in-place edit) they are re-packed and re-uploaded before the engine is handed out, so
        forward/rollouts never run stale weights."""
        if self._engine is None:
            self._check_supported()
            from .engine import RolloutEngine
            self._engine = RolloutEngine(self.state_dict(), self._engine_device, activation=self._activation_name())
            self._packed_fp = self._fingerprint()
        elif self._packed_fp is not None:
            fp = self._fingerprint()
            if fp != self._packed_fp:
                self._repack()
                self._packed_fp = fp
        return self._engine

    def _repack(self):
        """New parameter values -> the engine's weight image.  Parameters that live on the engine's device are
        concatenated there (blob order of weights.blob_layout) and packed by the device-side packer
        (phnn_update_weights_dev): an optimizer step then costs no device-to-host copy, host packing or upload.
        Anything else goes through the host packer."""
        sd = self.state_dict()
        eng = self._engine
        if hasattr(eng, "update_weights_dev") and all(t.device == eng.device for t in sd.values()):
            blob = torch.cat([sd[k].detach().reshape(-1).to(torch.float32) for k, _, _ in eng.layout])
            eng.update_weights_dev(blob)
        else:
            eng.update_weights(sd)

    def _activation_name(self):
        for mod in self.modules():
            if isinstance(mod, MLP):
                return mod.activation_name
        return "tanh"

    # Tanh: every kernel family; SiLU / ReLU / ELU (alpha = 1) / GELU (erf form): the all-f32 rollout kernels
    SUPPORTED_ACTIVATIONS = ("Tanh", "SiLU", "ReLU", "ELU", "GELU")

    def set_engine(self, engine):
        """Attach an already built engine (tests use this to run the host logic without a GPU)."""
        self._engine = engine
        self._packed_fp = None  # externally managed: no automatic re-packing
        return self

    def refresh_engine(self):
        """Call after changing parameters in place; the next call re-packs and re-uploads them."""
        if self._engine is not None and hasattr(self._engine, "close"):
            self._engine.close()
        self._engine = None

    def use_device(self, device):
        self._engine_device = str(device)
        self.refresh_engine()
        return self

    def load_state_dict(self, state_dict, *a, **k):
        r = super().load_state_dict(state_dict, *a, **k)
        self.refresh_engine()
        return r

    def _check_supported(self):
        for name, mod in self.named_modules():
            if isinstance(mod, MassMatrixNetwork) and mod.activation_name != "Tanh":
                raise NotImplementedError(f"{name}: the M_net.mlp kernel implements Tanh, got {mod.activation_name}")
            if isinstance(mod, MLP) and (not mod.plain or mod.activation_name not in self.SUPPORTED_ACTIVATIONS):
                raise NotImplementedError(
                    f"{name}: the rollout kernels implement Linear + Tanh / SiLU / ReLU / ELU / GELU MLPs (bias, no LayerNorm/Dropout); "
                    f"got activation={mod.activation_name}, plain={mod.plain}")
        acts = {mod.activation_name for mod in self.modules() if isinstance(mod, MLP)}
        if len(acts) > 1:
            raise NotImplementedError(f"the kernels take ONE activation per model; this one mixes {sorted(acts)}")

    @staticmethod
    def _flatten(x):  # src/pHNN.py:58-66
        if x.ndim == 1:
            return x.unsqueeze(0)
        if x.ndim > 2:
            return x.reshape(-1, x.shape[-1])
        return x


class pHNN(_EngineBacked):
    def __init__(self, config_path: str):
        super().__init__()
        with open(config_path, "r") as f:
            config = yaml.safe_load(f)
        mc = config["model"]
        n, m = mc["state_dim"], mc["input_dim"]
        self.J = nn.Parameter(torch.randn(n, n))
        self.R_net = _mlp_from(mc["R_mlp"], n, n * n)
        self.H_net = _mlp_from(mc["H_mlp"], n, 1)
        if mc.get("fixed_G", False):
            self.register_buffer("G_fixed", torch.tensor(mc["G_value"], dtype=torch.float32))
            self.G_net = None
        else:
            self.G_net = _mlp_from(mc["G_mlp"], n, m * n)

    def forward(self, x, u):
        x, u = self._flatten(x), self._flatten(u)
        return call_model_fn(self, x, u)


class CartPoleMassMatrix(nn.Module):
    """Parameters of M(theta) = [[a, b cos], [b cos, c]] (src/mass_matrix.py:239-370).  forward/inverse are the
    small closed forms, kept for the reference's diagnostic helpers; the hot path evaluates them in-kernel."""

    def __init__(self, init_a=1.0, init_b=0.1, init_c=1.0):
        super().__init__()
        self.log_a = nn.Parameter(torch.log(torch.tensor(init_a)))
        self.b = nn.Parameter(torch.tensor(init_b))
        self.log_c = nn.Parameter(torch.log(torch.tensor(init_c)))

    def _abc(self):
        return (torch.exp(self.log_a) + 1e-3).item(), self.b.item(), (torch.exp(self.log_c) + 1e-3).item()

    def forward(self, q):
        a, b, c = self._abc()
        bc = b * torch.cos(q[:, 1])
        M = torch.zeros(q.shape[0], 2, 2, dtype=q.dtype, device=q.device)
        M[:, 0, 0], M[:, 1, 1] = a, c
        M[:, 0, 1] = M[:, 1, 0] = bc
        return M

    def inverse(self, q):
        a, b, c = self._abc()
        bc = b * torch.cos(q[:, 1])
        det = a * c - bc ** 2 + 1e-6
        Mi = torch.zeros(q.shape[0], 2, 2, dtype=q.dtype, device=q.device)
        Mi[:, 0, 0], Mi[:, 1, 1] = c / det, a / det
        Mi[:, 0, 1] = Mi[:, 1, 0] = -bc / det
        return Mi

    def get_parameters_dict(self):
        a, b, c = self._abc()
        return {"a": a, "b": b, "c": c}


class MassMatrixNetwork(nn.Module):
    """Parameters of the general learnable mass matrix (src/mass_matrix.py:15-216), same keys as the reference:
    'constant' -> L_tril (q_dim, q_dim); 'diagonal' / 'full' -> mlp = Sequential(Linear, act, ..., Linear) with q_dim
    resp. q_dim (q_dim + 1) / 2 outputs.  The hot path evaluates M(q) and M^-1(q) in-kernel (q_dim = 2); forward /
    inverse below are the small closed forms for the reference's host-side helpers."""

    def __init__(self, q_dim, mass_type="diagonal", hidden_sizes=(64, 64), activation=None, init_scale=1.0):
        super().__init__()
        if mass_type not in ("constant", "diagonal", "full"):
            raise ValueError(f"Unknown mass_type: {mass_type}")
        self.q_dim, self.mass_type, self.init_scale = q_dim, mass_type, init_scale
        self.activation_name = type(activation).__name__ if activation is not None else "Tanh"
        if mass_type == "constant":
            self.L_tril = nn.Parameter(torch.eye(q_dim) * init_scale)
            self.mlp = None
            return
        out = q_dim if mass_type == "diagonal" else q_dim * (q_dim + 1) // 2
        mods, prev = [], q_dim
        for h in hidden_sizes:
            mods += [nn.Linear(prev, h), activation if activation is not None else nn.Tanh()]
            prev = h
        mods.append(nn.Linear(prev, out))
        self.mlp = nn.Sequential(*mods)
        nn.init.zeros_(self.mlp[-1].weight)
        nn.init.zeros_(self.mlp[-1].bias)
        if mass_type == "full":  # diagonal Cholesky entries start at log(init_scale)
            with torch.no_grad():
                idx, k = [], 0
                for i in range(q_dim):
                    idx.append(k)
                    k += i + 2
                self.mlp[-1].bias[idx] = math.log(init_scale)

    def _chol(self, raw):
        """(..., n, n) raw lower-triangular entries -> L with softplus(diag) + 1e-3"""
        n = self.q_dim
        L = torch.tril(raw)
        d = torch.nn.functional.softplus(torch.diagonal(L, dim1=-2, dim2=-1)) + 1e-3
        return L - torch.diag_embed(torch.diagonal(L, dim1=-2, dim2=-1)) + torch.diag_embed(d)

    def forward(self, q):
        B = q.shape[0]
        if self.mass_type == "constant":
            L = self._chol(self.L_tril)
            return (L @ L.T).unsqueeze(0).expand(B, -1, -1)
        o = self.mlp(q)
        if self.mass_type == "diagonal":
            return torch.diag_embed(torch.exp(o) + 1e-3)
        raw = torch.zeros(B, self.q_dim, self.q_dim, dtype=o.dtype, device=o.device)
        r, c = torch.tril_indices(self.q_dim, self.q_dim, offset=0)
        raw[:, r, c] = o
        L = self._chol(raw)
        return torch.bmm(L, L.transpose(1, 2))

    def inverse(self, q):
        return torch.linalg.inv(self.forward(q))


class pHNN_Canonical(_EngineBacked):
    def __init__(self, config_path: str):
        super().__init__()
        with open(config_path, "r") as f:
            config = yaml.safe_load(f)
        mc = config["model"]
        self.state_dim, self.input_dim = mc["state_dim"], mc["input_dim"]
        self.q_dim = self.state_dim // 2
        mass = mc.get("mass_matrix", {})
        mass_type = mass.get("type", "cartpole")
        if mass_type == "cartpole":  # src/pHNN_canonical.py:67-86
            self.M_net = CartPoleMassMatrix(mass.get("init_a", 1.0), mass.get("init_b", 0.1), mass.get("init_c", 1.0))
        else:
            self.M_net = MassMatrixNetwork(self.q_dim, mass_type, mass.get("hidden_sizes", [64, 64]),
                                           _activation(mass.get("activation", "nn.Tanh"))(), mass.get("init_scale", 1.0))
        self.H_net = _mlp_from(mc["H_mlp"], self.state_dim, 1)
        J = torch.zeros(self.state_dim, self.state_dim)
        J[:self.q_dim, self.q_dim:] = torch.eye(self.q_dim)
        J[self.q_dim:, :self.q_dim] = -torch.eye(self.q_dim)
        self.register_buffer("J", J)
        self.R_diag_raw = nn.Parameter(torch.ones(self.state_dim) * 0.1)
        if not mc.get("fixed_G", False):
            raise ValueError("pHNN_Canonical requires fixed_G=True")
        self.register_buffer("G", torch.tensor(mc["G_value"], dtype=torch.float32))

    def get_R_matrix(self, batch_size):
        R = torch.diag(torch.nn.functional.softplus(self.R_diag_raw) + 1e-4)
        return R.unsqueeze(0).expand(batch_size, -1, -1)

    def forward(self, y, u, return_intermediate=False):
        """-> (dy, H, intermediate | None).  With return_intermediate the reference returns a dict of intermediates
        (src/pHNN_canonical.py:259-271); the training loop reads 'q_dot_reconstructed' from it
        (scripts/train_cartpole_phnn_canonical.py:141-146), which IS the first half of dy (q_dot = M^-1 p).  The
        keys that are views of the outputs or cheap functions of the inputs are provided; 'dH_dz' and 'dz_dt' live
        only inside the fused kernel and are not returned."""
        y, u = self._flatten(y), self._flatten(u)
        dy, H = call_model_fn(self, y, u)
        if not return_intermediate:
            return dy, H, None
        q, qd = y[:, :self.q_dim], y[:, self.q_dim:]
        p = torch.bmm(self.M_net(q), qd.unsqueeze(-1)).squeeze(-1)
        inter = {"z": torch.cat([q, p], dim=1), "q": q, "p": p, "q_dot_reconstructed": dy[:, :self.q_dim],
                 "R": self.get_R_matrix(y.shape[0])}
        return dy, H, inter

    def get_velocity_reconstruction(self, y):
        q, qd = y[:, :self.q_dim], y[:, self.q_dim:]
        p = torch.bmm(self.M_net(q), qd.unsqueeze(-1))
        return torch.bmm(self.M_net.inverse(q), p).squeeze(-1)


class ODEFunc(_EngineBacked):
    def __init__(self, state_dim=4, action_dim=1, hidden_sizes=[128, 128, 128], activation="tanh", layer_norm=False):
        super().__init__()
        self.state_dim, self.action_dim = state_dim, action_dim
        self.current_action = None
        acts = {"relu": nn.ReLU, "tanh": nn.Tanh, "elu": nn.ELU, "gelu": nn.GELU}
        if activation not in acts:
            raise ValueError(f"Unknown activation: {activation}")
        self._plain = not layer_norm
        self._act = activation
        mods, prev = [], state_dim + action_dim
        for h in hidden_sizes:
            mods.append(nn.Linear(prev, h))
            if layer_norm:
                mods.append(nn.LayerNorm(h))
            mods.append(acts[activation]())
            prev = h
        mods.append(nn.Linear(prev, state_dim))
        self.network = nn.Sequential(*mods)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.zeros_(m.bias)

    def _check_supported(self):
        if not self._plain:
            raise NotImplementedError("the ODEFunc kernels implement tanh (default), relu, elu and gelu, without LayerNorm")

    def _activation_name(self):
        return self._act

    def forward(self, t, state):
        if self.current_action is None:
            raise RuntimeError("current_action must be set before calling forward")
        action = self.current_action
        if action.shape[0] == 1 and state.shape[0] > 1:
            action = action.expand(state.shape[0], -1)
        dx, _ = call_model_fn(self, state, action)
        return dx

    def dynamics(self, y, u):
        """model(y,u) -> (dy, H=0) view used by the integrators (no side channel)."""
        return call_model_fn(self, self._flatten(y), self._flatten(u))
