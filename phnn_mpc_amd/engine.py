"""RolloutEngine: the thin Python face of the C-ABI (include/phnn_mpc.h) on torch device tensors.

torch is plumbing here (device memory, streams); all arithmetic happens in the gfx950 kernels of
csrc/libphnn_mpc.so.  There is no fallback: without the library or without a GPU, construction raises.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _capi, weights


class PhnnError(RuntimeError):
    pass


def _check(lib, handle, rc):
    if rc != 0:
        msg = lib.phnn_last_error(handle)
        msg = msg.decode() if msg else ""
        if msg.startswith("Unknown integrator"):
            raise ValueError(msg)  # the reference raises ValueError (src/integrators.py:172,226)
        raise PhnnError(f"phnn_mpc error {rc}: {msg}")


class RolloutEngine:
    """One dynamics model resident on one GPU.

    state_dict: the reference's state_dict (torch tensors or numpy arrays), or a checkpoint wrapping it.
    """

    def __init__(self, state_dict, device="cuda:0", kind=None, activation="tanh", matmul=None, force_matmul=False,
                 max_waves=None, split="auto"):
        """matmul: 'default' | 'f32' | 'bf16x3' | 'f16x2' (None: the PHNN_MATMUL environment variable, else
        'default').  The environment is read HERE, on the Python side, as a default only; the C-ABI takes the
        explicit phnn_options.  f16x2 on a model narrower than 128 needs force_matmul=True (known to exceed the
        stated tolerance there).  max_waves (None: PHNN_MAX_WAVES, else 8): waves per workgroup cap.
        split: 'auto' | 'never' | 'always' -- the small-batch kernels that put four waves on every 16-rollout tile
        (bitwise the same results; automatic while the batch has at most two tiles per CU)."""
        self.lib = _capi.load_library()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise PhnnError("RolloutEngine needs a GPU device (cuda:N); there is no CPU path")
        if not torch.cuda.is_available():
            raise PhnnError("no GPU visible to torch; the rollout engine has no CPU fallback")
        self.activation = activation
        self.desc, self.blob = weights.pack_state_dict(state_dict, kind=kind, activation=activation)
        self.n, self.m, self.kind = self.desc.n, self.desc.m, self.desc.kind
        self.layout = weights.blob_layout(state_dict, kind=self.kind)  # [(state_dict key, offset, shape)] of the blob
        self._wg_ws = None
        self._tape_gen, self._tape_token = 0, None  # tapes kept by the last rollout_trajectory(tapes=True), if still valid
        self.use_tapes = os.environ.get("PHNN_NO_TAPES", "0") != "1"
        if matmul is None:
            matmul = os.environ.get("PHNN_MATMUL", "default")
            force_matmul = force_matmul or "PHNN_MATMUL" in os.environ  # an explicit environment override is a force
        if matmul not in _capi.MATMUL_MODES:
            raise ValueError(f"matmul must be one of {sorted(_capi.MATMUL_MODES)}, got {matmul!r}")
        if max_waves is None:
            max_waves = int(os.environ.get("PHNN_MAX_WAVES", "0"))
        self.options = _capi.Options()
        self.options.matmul_mode = _capi.MATMUL_MODES[matmul]
        self.options.force_matmul = int(bool(force_matmul))
        self.options.max_waves = int(max_waves)
        if split not in _capi.SPLIT_MODES:
            raise ValueError(f"split must be one of {sorted(_capi.SPLIT_MODES)}, got {split!r}")
        self.options.split_tiles = _capi.SPLIT_MODES[split]
        # K1 -> K2 activation stash (Euler): on unless PHNN_NO_STASH=1; capped so a huge batch falls back to
        # the recompute kernels instead of allocating more than max_stash_bytes of HBM
        self.use_stash = os.environ.get("PHNN_NO_STASH", "0") != "1"
        self.max_stash_bytes = int(float(os.environ.get("PHNN_MAX_STASH_GB", "96")) * (1 << 30))
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        h = C.c_void_p()
        rc = self.lib.phnn_create_ex(C.byref(self.desc), self.blob.ctypes.data_as(C.POINTER(C.c_float)),
                                     self.blob.size, idx, C.byref(self.options), C.byref(h))
        _check(self.lib, None, rc)
        self.h = h

    def update_weights(self, state_dict):
        """Re-pack and re-upload the weights of the same architecture (after an optimizer step / load_state_dict)."""
        desc, blob = weights.pack_state_dict(state_dict, kind=self.kind, activation=self.activation)
        if blob.size != self.blob.size:
            raise PhnnError("update_weights: the state_dict describes another architecture")
        self.blob = blob
        self._tape_token = None  # tapes of the old weights
        rc = self.lib.phnn_update_weights(self.h, blob.ctypes.data_as(C.POINTER(C.c_float)), blob.size, self._stream())
        _check(self.lib, self.h, rc)

    def update_weights_dev(self, blob_dev):
        """The same from a float32 blob on the engine's device (weights.pack_state_dict order, e.g. torch.cat of the
        module's parameters and buffers): padded and packed by one small kernel in stream order -- no device-to-host copy,
        host packing or upload (phnn_update_weights_dev)."""
        if blob_dev.device != self.device or blob_dev.dtype != torch.float32 or not blob_dev.is_contiguous():
            raise PhnnError("update_weights_dev: a contiguous float32 tensor on the engine's device is required")
        if blob_dev.numel() != self.blob.size:
            raise PhnnError("update_weights_dev: the blob describes another architecture")
        self._tape_token = None  # tapes of the old weights
        rc = self.lib.phnn_update_weights_dev(self.h, blob_dev.data_ptr(), blob_dev.numel(), self._stream())
        _check(self.lib, self.h, rc)

    def read_image(self):
        """The packed weight image as the kernels stage it (tests: host-packed vs device-packed)."""
        n = C.c_size_t()
        _check(self.lib, self.h, self.lib.phnn_read_image(self.h, None, 0, C.byref(n), None))
        out = np.empty(n.value, np.float32)
        rc = self.lib.phnn_read_image(self.h, out.ctypes.data_as(C.POINTER(C.c_float)), out.size, None, self._stream())
        _check(self.lib, self.h, rc)
        return out

    def close(self):
        if getattr(self, "h", None):
            self.lib.phnn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _t(self, a, shape=None):
        t = torch.as_tensor(a, dtype=torch.float32, device=self.device)
        if shape is not None:
            if t.numel() == 0:  # an empty batch: -1 is ambiguous for reshape
                shape = tuple(0 if d == -1 else d for d in shape)
            t = t.reshape(shape)
        return t.contiguous()

    def _controls(self, u, B):
        """(B,H,m) view of the controls; H is read from a 3-D input so that B = 0 keeps its horizon."""
        u = self._t(u)
        H = u.shape[1] if u.dim() == 3 else (u.numel() // max(B * self.m, 1))
        return u.reshape(B, H, self.m), H

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None

    def _integ(self, integrator):
        if isinstance(integrator, str):
            if integrator not in _capi.INTEGRATORS:
                raise ValueError(f"Unknown integrator: {integrator}")
            return _capi.INTEGRATORS[integrator]
        return int(integrator)

    # ------------------------------------------------------------------ model(x,u), VJP
    def forward(self, x, u):
        x = self._t(x, (-1, self.n))
        u = self._t(u, (-1, self.m))
        B = x.shape[0]
        dx = torch.empty_like(x)
        H = torch.empty(B, dtype=torch.float32, device=self.device)
        _check(self.lib, self.h, self.lib.phnn_model_forward(self.h, self._p(x), self._p(u), B, self._p(dx), self._p(H),
                                                             self._stream()))
        return dx, H

    def vjp(self, x, u, lam):
        x = self._t(x, (-1, self.n))
        u = self._t(u, (-1, self.m))
        lam = self._t(lam, (-1, self.n))
        B = x.shape[0]
        xb = torch.empty_like(x)
        ub = torch.empty(B, self.m, dtype=torch.float32, device=self.device)
        _check(self.lib, self.h, self.lib.phnn_model_vjp(self.h, self._p(x), self._p(u), self._p(lam), B, self._p(xb),
                                                         self._p(ub), self._stream()))
        return xb, ub

    # ------------------------------------------------------------------ rollouts
    def rollout_cost(self, x0, u, cost, integrator="euler", dt=0.02, want_traj=False, traj_out=None):
        """K1.  x0 (B,n), u (B,H,m) -> cost (B) [, traj (B,H+1,n)]."""
        x0 = self._t(x0, (-1, self.n))
        B = x0.shape[0]
        u, H = self._controls(u, B)
        c = torch.empty(B, dtype=torch.float32, device=self.device)
        traj = traj_out
        if traj is None and want_traj:
            traj = torch.empty(B, H + 1, self.n, dtype=torch.float32, device=self.device)
        rc = self.lib.phnn_rollout_fwd(self.h, self._p(x0), self._p(u), B, H, C.byref(cost), self._integ(integrator),
                                       float(dt), self._p(c), self._p(traj), None, self._stream())
        _check(self.lib, self.h, rc)
        return (c, traj) if (want_traj or traj_out is not None) else c

    def workspace_bytes(self, B, H, integrator="euler"):
        return int(self.lib.phnn_workspace_bytes(self.h, int(B), int(H), self._integ(integrator)))

    def rollout_cost_grad(self, x0, u, cost, integrator="euler", dt=0.02, want_grad_x0=False, workspace=None,
                          after_forward=None):
        """K1 + K2.  -> (cost (B), grad_u (B,H,m)[, grad_x0 (B,n)]).  `workspace`: optional dict reused across
        calls to avoid re-allocating the trajectory / outputs.  `after_forward(cost)` is called once K1 is enqueued
        and before K2 is: the costs are final then, so a collective on them overlaps the adjoint kernel."""
        x0 = self._t(x0, (-1, self.n))
        B = x0.shape[0]
        u, H = self._controls(u, B)
        ws = workspace if workspace is not None else {}
        integ = self._integ(integrator)
        key = (B, H, integ)
        if ws.get("key") != key:
            ws["key"] = key
            nbytes = self.workspace_bytes(B, H, integ) if self.use_stash else 0
            ws["stash"] = (torch.empty(nbytes, dtype=torch.uint8, device=self.device)
                           if 0 < nbytes <= self.max_stash_bytes else None)
            ws["traj"] = torch.empty(B, H + 1, self.n, dtype=torch.float32, device=self.device)
            ws["cost"] = torch.empty(B, dtype=torch.float32, device=self.device)
            ws["grad_u"] = torch.empty(B, H, self.m, dtype=torch.float32, device=self.device)
            ws["grad_x0"] = torch.empty(B, self.n, dtype=torch.float32, device=self.device)
        st = self._stream()
        stash = self._p(ws["stash"])
        rc = self.lib.phnn_rollout_fwd(self.h, self._p(x0), self._p(u), B, H, C.byref(cost), integ, float(dt),
                                       self._p(ws["cost"]), self._p(ws["traj"]), stash, st)
        _check(self.lib, self.h, rc)
        if after_forward is not None:
            after_forward(ws["cost"])
        rc = self.lib.phnn_rollout_grad(self.h, self._p(x0), self._p(u), B, H, C.byref(cost), integ, float(dt),
                                        self._p(ws["traj"]), stash, self._p(ws["grad_u"]),
                                        self._p(ws["grad_x0"]) if want_grad_x0 else None, st)
        _check(self.lib, self.h, rc)
        if want_grad_x0:
            return ws["cost"], ws["grad_u"], ws["grad_x0"]
        return ws["cost"], ws["grad_u"]

    def rollout_vjp(self, x0, u, traj, cost, integrator="euler", dt=0.02, traj_bar=None, cost_bar=None):
        """General reverse pass: cotangents on the trajectory (B,H+1,n) and/or the cost (B) -> (grad_u, grad_x0)."""
        x0 = self._t(x0, (-1, self.n))
        B = x0.shape[0]
        u, H = self._controls(u, B)
        traj = self._t(traj, (B, H + 1, self.n))
        tb = self._t(traj_bar, (B, H + 1, self.n)) if traj_bar is not None else None
        cb = self._t(cost_bar, (B,)) if cost_bar is not None else None
        gu = torch.empty(B, H, self.m, dtype=torch.float32, device=self.device)
        gx = torch.empty(B, self.n, dtype=torch.float32, device=self.device)
        rc = self.lib.phnn_rollout_vjp(self.h, self._p(x0), self._p(u), B, H, C.byref(cost), self._integ(integrator),
                                       float(dt), self._p(traj), None, self._p(tb), self._p(cb), self._p(gu),
                                       self._p(gx), self._stream())
        _check(self.lib, self.h, rc)
        return gu, gx

    # ------------------------------------------------------------------ training side: parameter gradients (row f4)
    @property
    def has_wgrad(self):
        """True when the handle has weight-gradient kernels (pHNN and canonical pHNN variants)."""
        return self.lib.phnn_wgrad_workspace_bytes(self.h, 16, 1, 0) > 0

    def _wgrad_workspace(self, B, H, integ):
        need = int(self.lib.phnn_wgrad_workspace_bytes(self.h, int(B), int(H), int(integ)))
        if need == 0 and B > 0:
            raise PhnnError("this model variant has no weight-gradient kernels (pHNN and canonical pHNN have them)")
        if self._wg_ws is None or self._wg_ws.numel() < need:
            self._wg_ws = torch.empty(max(need, 1), dtype=torch.uint8, device=self.device)
            self._tape_token = None  # a new buffer: whatever tapes the old one held are gone
        return self._wg_ws

    def rollout_trajectory(self, x0, u, integrator="euler", dt=0.02, want_dx=False, tapes=False):
        """Training rollout (no clamp, no cost): x0 (B,n), u (B,H,m) -> traj (B,H+1,n) [, dX (B,H,n) = f(x_t,u_t)].
        tapes=True (models with weight-gradient kernels): K1 also keeps the tapes of every dynamics evaluation in the
        weight-gradient workspace; `self.tape_token` then identifies them, and rollout_wgrad(..., tape_token=that)
        uses them as long as nothing else has touched the workspace or the weights since."""
        x0 = self._t(x0, (-1, self.n))
        B = x0.shape[0]
        u, H = self._controls(u, B)
        integ = self._integ(integrator)
        traj = torch.empty(B, H + 1, self.n, dtype=torch.float32, device=self.device)
        dX = torch.empty(B, H, self.n, dtype=torch.float32, device=self.device) if want_dx else None
        ws = None
        if tapes and self.use_tapes and B > 0 and self.has_wgrad:
            ws = self._wgrad_workspace(B, H, integ)
            self._tape_gen += 1
            self._tape_token = (self._tape_gen, B, H, integ, float(dt))
        rc = self.lib.phnn_rollout_trajectory_ws(self.h, self._p(x0), self._p(u), B, H, integ, float(dt), self._p(traj),
                                                 self._p(dX), self._p(ws), self._stream())
        _check(self.lib, self.h, rc)
        return (traj, dX) if want_dx else traj

    @property
    def tape_token(self):
        """Token of the tapes the weight-gradient workspace currently holds (None: none / invalidated)."""
        return self._tape_token

    def rollout_wgrad(self, x0, u, traj, integrator="euler", dt=0.02, traj_bar=None, dx_bar=None, grad_theta=None,
                      accumulate=False, tape_token=None):
        """Reverse pass of a training rollout: cotangents on the trajectory (B,H+1,n) and on the per-step derivatives
        (B,H,n) -> (grad_theta (P,) in weight-blob layout, grad_u (B,H,m), grad_x0 (B,n)).  tape_token: the token
        rollout_trajectory(tapes=True) left for this very rollout; used only if the tapes are still the current ones
        (otherwise the adjoint recomputes the forward pass, same results to rounding)."""
        x0 = self._t(x0, (-1, self.n))
        B = x0.shape[0]
        u, H = self._controls(u, B)
        integ = self._integ(integrator)
        traj = self._t(traj, (B, H + 1, self.n))
        tb = self._t(traj_bar, (B, H + 1, self.n)) if traj_bar is not None else None
        db = self._t(dx_bar, (B, H, self.n)) if dx_bar is not None else None
        if grad_theta is None:
            grad_theta = torch.empty(self.blob.size, dtype=torch.float32, device=self.device)
            accumulate = False
        gu = torch.empty(B, H, self.m, dtype=torch.float32, device=self.device)
        gx = torch.empty(B, self.n, dtype=torch.float32, device=self.device)
        ws = self._wgrad_workspace(B, H, integ)
        use_tapes = (tape_token is not None and tape_token == self._tape_token
                     and tape_token[1:] == (B, H, integ, float(dt)))
        if not use_tapes:
            # records and slab of THIS shape start at offset 0 of the shared workspace and may reach into the tape region
            # of whatever (smaller / other-shaped) rollout left its tapes there: those tapes are no longer trustworthy
            self._tape_token = None
        flags = (_capi.WGRAD_ACCUMULATE if accumulate else 0) | (_capi.WGRAD_TAPES if use_tapes else 0)
        rc = self.lib.phnn_rollout_wgrad(self.h, self._p(x0), self._p(u), B, H, integ, float(dt), self._p(traj),
                                         self._p(tb), self._p(db), self._p(ws), self._p(grad_theta), flags,
                                         self._p(gu), self._p(gx), self._stream())
        _check(self.lib, self.h, rc)
        return grad_theta, gu, gx

    def model_wgrad(self, x, u, lam, Hbar=None, grad_theta=None, accumulate=False):
        """Single evaluations: -> (grad_theta (P,) of sum lam.f + Hbar H, xbar (N,n), ubar (N,m))."""
        x = self._t(x, (-1, self.n))
        u = self._t(u, (-1, self.m))
        lam = self._t(lam, (-1, self.n))
        N = x.shape[0]
        hb = self._t(Hbar, (N,)) if Hbar is not None else None
        if grad_theta is None:
            grad_theta = torch.empty(self.blob.size, dtype=torch.float32, device=self.device)
            accumulate = False
        xb = torch.empty_like(x)
        ub = torch.empty(N, self.m, dtype=torch.float32, device=self.device)
        ws = self._wgrad_workspace(N, 0, 0)
        self._tape_token = None  # point-mode records may reach into the tape region of a rollout-sized workspace
        rc = self.lib.phnn_model_wgrad(self.h, self._p(x), self._p(u), self._p(lam), self._p(hb), N, self._p(ws),
                                       self._p(grad_theta), int(accumulate), self._p(xb), self._p(ub), self._stream())
        _check(self.lib, self.h, rc)
        return grad_theta, xb, ub

    def mass_cotangents(self, B, H=0, integrator="euler"):
        """(q (P,2), Mbar (P,2,2)) of the evaluation points of the LAST rollout_wgrad(B, H, integrator) / model_wgrad(B
        points, H = 0) call, read from the records in the weight-gradient workspace -- canonical models with a
        MassMatrixNetwork: the cotangent of M(q) at every evaluation (phnn_wgrad_record_info).  P = 16 * records: points
        beyond the batch carry zero cotangents."""
        rf, so, ss = C.c_int32(), C.c_int32(), C.c_int32()
        _check(self.lib, self.h, self.lib.phnn_wgrad_record_info(self.h, C.byref(rf), C.byref(so), C.byref(ss)))
        tiles = (int(B) + 15) // 16
        n_rec = tiles if H <= 0 else tiles * int(H) * (4 if self._integ(integrator) == 1 else 1)
        rec = self._wg_ws.view(torch.float32)[: n_rec * rf.value].view(n_rec, rf.value)
        sm = rec[:, so.value:].reshape(n_rec * 16, ss.value)
        return sm[:, 24:26], sm[:, 20:24].reshape(-1, 2, 2)

    def named_grads(self, grad_theta):
        """{state_dict key: tensor view of the parameter's shape} of a gradient blob."""
        return weights.unpack_grad_blob(None, grad_theta, layout=self.layout)

    # ------------------------------------------------------------------ Adam on the controls (K3)
    def adam_step(self, u, grad, exp_avg, exp_avg_sq, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, cost=None,
                  best_cost=None, best_u=None, u_min=None, u_max=None):
        """In-place torch.optim.Adam step on u (B,H,m) (+ optional best-iterate tracking)."""
        for t in (u, grad, exp_avg, exp_avg_sq):
            assert t.is_contiguous() and t.dtype == torch.float32 and t.device == self.device
        per = u[0].numel() if u.dim() > 1 else 1
        has_b = u_min is not None and u_max is not None
        rc = self.lib.phnn_adam_step(self.h, self._p(u), self._p(grad), self._p(exp_avg), self._p(exp_avg_sq),
                                     u.numel(), float(lr), float(beta1), float(beta2), float(eps), int(step),
                                     self._p(cost), self._p(best_cost), self._p(best_u), per,
                                     float(u_min) if has_b else 0.0, float(u_max) if has_b else 0.0, int(has_b),
                                     self._stream())
        _check(self.lib, self.h, rc)

    # ------------------------------------------------------------------ the whole shooting solve (K1, K2, K3 x iters)
    def solve(self, x0, u_init, cost, integrator="euler", dt=0.02, lr=0.015, iters=30, track_best=False, record_costs=True,
              beta1=0.9, beta2=0.999, eps=1e-8, workspace=None):
        """phnn_solve: Adam on the control sequences of B independent problems, the loops of
        src/mpc_controller.py:164-209 / src/mpc_controller_canonical.py:163-228, as ONE library call that enqueues the
        K1 / K2 / K3 launches of every iteration (no Python between them).  -> dict as solver.shooting_solve, same
        results bit for bit."""
        x0 = self._t(x0, (-1, self.n))
        B = x0.shape[0]
        u_init, H = self._controls(u_init, B)
        integ = self._integ(integrator)
        ws = workspace if workspace is not None else {}
        key = ("solve", B, H, integ, int(iters))
        if ws.get("skey") != key:
            ws["skey"] = key
            nbytes = self.workspace_bytes(B, H, integ) if self.use_stash else 0
            ws["s_stash"] = (torch.empty(nbytes, dtype=torch.uint8, device=self.device)
                             if 0 < nbytes <= self.max_stash_bytes else None)
            f = dict(dtype=torch.float32, device=self.device)
            ws["s_traj"], ws["s_cost"] = torch.empty(B, H + 1, self.n, **f), torch.empty(B, **f)
            ws["s_grad"], ws["s_m"], ws["s_v"] = (torch.empty(B, H, self.m, **f) for _ in range(3))
        u = u_init.detach().clone().contiguous()
        f = dict(dtype=torch.float32, device=self.device)
        costs = torch.empty(int(iters), B, **f) if record_costs else None
        best_cost = torch.empty(B, **f) if track_best else None
        best_u = torch.empty(B, H, self.m, **f) if track_best else None
        opt = _capi.SolveOptions(int(iters), float(lr), float(beta1), float(beta2), float(eps), int(bool(track_best)))
        rc = self.lib.phnn_solve(self.h, self._p(x0), self._p(u), B, H, C.byref(cost), integ, float(dt), C.byref(opt),
                                 self._p(ws["s_m"]), self._p(ws["s_v"]), self._p(ws["s_grad"]), self._p(ws["s_cost"]),
                                 self._p(ws["s_traj"]), self._p(ws["s_stash"]), self._p(costs), self._p(best_cost),
                                 self._p(best_u), self._stream())
        _check(self.lib, self.h, rc)
        out = {"u_last": u, "costs": costs}
        if track_best:
            out["best_u"], out["best_cost"] = best_u, best_cost
        return out

    # ------------------------------------------------------------------ the plant, on the device
    def plant_step(self, plant, state, action, action_stride, u_min=None, u_max=None, state_f32=None, done_step=None,
                   step=0, step_dev=None, log_states=None, log_controls=None):
        """One float64 Euler step of the reference cart-pole for B plants, in place on `state` (B,4) float64.
        action: float32 tensor, plant b takes action.view(-1)[b * action_stride].  See include/phnn_mpc.h."""
        assert state.dtype == torch.float64 and state.is_contiguous() and state.device == self.device
        assert action.dtype == torch.float32 and action.is_contiguous() and action.device == self.device
        B = state.shape[0]
        has_b = u_min is not None and u_max is not None
        rc = self.lib.phnn_plant_step(self.h, C.byref(plant), self._p(state), self._p(action), int(action_stride), B,
                                      int(has_b), float(u_min) if has_b else 0.0, float(u_max) if has_b else 0.0,
                                      self._p(state_f32), self._p(done_step), self._p(step_dev), int(step),
                                      self._p(log_states), self._p(log_controls), self._stream())
        _check(self.lib, self.h, rc)

    def shift_controls(self, src, dst, step_dev=None):
        """dst[b,t] = src[b,t+1], dst[b,H-1] = 0 (warm start of the next solve); advances *step_dev if given."""
        B, H, m = src.shape
        rc = self.lib.phnn_shift_controls(self.h, self._p(src), self._p(dst), B, H, m, self._p(step_dev), self._stream())
        _check(self.lib, self.h, rc)

    def advance_step(self, step_dev):
        rc = self.lib.phnn_shift_controls(self.h, None, None, 0, 1, 1, self._p(step_dev), self._stream())
        _check(self.lib, self.h, rc)

    @property
    def variant(self):
        """e.g. 'phnn<n=4,hid=128,fixedG,f16x2>'"""
        return self.lib.phnn_variant_name(self.h).decode()

    @property
    def matmul_mode(self):
        v = self.variant
        return "f16x2" if "f16x2" in v else ("bf16x3" if "bf16x3" in v else "f32")

    def kernel_info(self, B, integrator="euler"):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self.lib.phnn_kernel_info(self.h, self._integ(integrator), C.byref(a), C.byref(b), C.byref(c), int(B))
        return {"rollouts_per_workgroup": a.value, "lds_bytes": b.value, "workgroups": c.value}


def as_numpy(t):
    return t.detach().cpu().numpy().astype(np.float64)
