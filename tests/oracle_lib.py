"""ctypes loader of oracle/libphnn_oracle.so (the CPU restatement; test infrastructure only).

Builds the library with `make -C oracle` when it is missing.  Nothing under phnn_mpc_amd/ imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from phnn_mpc_amd import _capi, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")
_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "libphnn_oracle.so")
        src = os.path.join(ORACLE_DIR, "phnn_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
        _lib = C.CDLL(so)
        for suf in ("f32", "f64"):
            getattr(_lib, f"oracle_create_{suf}").restype = C.c_void_p
            getattr(_lib, f"oracle_create_{suf}").argtypes = [C.POINTER(_capi.Desc), C.c_void_p, C.c_size_t]
            getattr(_lib, f"oracle_destroy_{suf}").argtypes = [C.c_void_p]
            getattr(_lib, f"oracle_forward_{suf}").argtypes = [C.c_void_p] * 3 + [C.c_long] + [C.c_void_p] * 2
            getattr(_lib, f"oracle_vjp_{suf}").argtypes = [C.c_void_p] * 4 + [C.c_long] + [C.c_void_p] * 2
            getattr(_lib, f"oracle_rollout_{suf}").argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int,
                                                               C.POINTER(_capi.Cost), C.c_int, C.c_double,
                                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
            getattr(_lib, f"oracle_rollout_vjp_{suf}").argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int,
                                                                   C.POINTER(_capi.Cost), C.c_int, C.c_double,
                                                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            getattr(_lib, f"oracle_wgrad_{suf}").argtypes = [C.c_void_p] * 5 + [C.c_long, C.c_void_p]
            getattr(_lib, f"oracle_rollout_wgrad_{suf}").argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int,
                                                                     C.c_int, C.c_double] + [C.c_void_p] * 7
            getattr(_lib, f"oracle_adam_{suf}").argtypes = [C.c_void_p] * 4 + [C.c_long] + [C.c_double] * 4 + [C.c_int]
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleModel:
    """One reference model restated on the CPU, in float32 ('f32') or float64 ('f64')."""

    def __init__(self, state_dict, precision="f64", kind=None, activation="tanh"):
        self.desc, self.blob = weights.pack_state_dict(state_dict, kind=kind, activation=activation)
        self.suf = precision
        self.dtype = np.float64 if precision == "f64" else np.float32
        self.n, self.m = self.desc.n, self.desc.m
        self.h = getattr(lib(), f"oracle_create_{precision}")(C.byref(self.desc), _ptr(self.blob), self.blob.size)
        if not self.h:
            raise RuntimeError("oracle_create failed (bad description / blob size)")

    def __del__(self):
        if getattr(self, "h", None):
            getattr(lib(), f"oracle_destroy_{self.suf}")(self.h)
            self.h = None

    def _a(self, x, shape=None):
        a = np.ascontiguousarray(np.asarray(x, dtype=self.dtype))
        return a if shape is None else a.reshape(shape)

    def forward(self, x, u):
        x, u = self._a(x, (-1, self.n)), self._a(u, (-1, self.m))
        B = x.shape[0]
        dx, H = np.empty((B, self.n), self.dtype), np.empty(B, self.dtype)
        getattr(lib(), f"oracle_forward_{self.suf}")(self.h, _ptr(x), _ptr(u), B, _ptr(dx), _ptr(H))
        return dx, H

    def vjp(self, x, u, lam):
        x, u, lam = self._a(x, (-1, self.n)), self._a(u, (-1, self.m)), self._a(lam, (-1, self.n))
        B = x.shape[0]
        xb, ub = np.empty((B, self.n), self.dtype), np.empty((B, self.m), self.dtype)
        getattr(lib(), f"oracle_vjp_{self.suf}")(self.h, _ptr(x), _ptr(u), _ptr(lam), B, _ptr(xb), _ptr(ub))
        return xb, ub

    def rollout(self, x0, U, cost, integrator, dt, grad=True, traj=True, nthreads=1):
        x0 = self._a(x0, (-1, self.n))
        B = x0.shape[0]
        U = self._a(U).reshape(B, -1, self.m)
        H = U.shape[1]
        c = np.empty(B, self.dtype)
        tr = np.empty((B, H + 1, self.n), self.dtype) if traj else None
        gu = np.empty((B, H, self.m), self.dtype) if grad else None
        gx = np.empty((B, self.n), self.dtype) if grad else None
        integ = _capi.INTEGRATORS[integrator] if isinstance(integrator, str) else int(integrator)
        getattr(lib(), f"oracle_rollout_{self.suf}")(self.h, _ptr(x0), _ptr(U), B, H, C.byref(cost), integ, float(dt),
                                                     _ptr(c), _ptr(tr), _ptr(gu), _ptr(gx), int(nthreads))
        return {"cost": c, "traj": tr, "grad_u": gu, "grad_x0": gx}

    def rollout_vjp(self, x0, U, cost, integrator, dt, traj_bar=None, cost_bar=None):
        x0 = self._a(x0, (-1, self.n))
        B = x0.shape[0]
        U = self._a(U).reshape(B, -1, self.m)
        H = U.shape[1]
        tb = None if traj_bar is None else self._a(traj_bar, (B, H + 1, self.n))
        cb = None if cost_bar is None else self._a(cost_bar, (B,))
        gu, gx = np.empty((B, H, self.m), self.dtype), np.empty((B, self.n), self.dtype)
        integ = _capi.INTEGRATORS[integrator] if isinstance(integrator, str) else int(integrator)
        getattr(lib(), f"oracle_rollout_vjp_{self.suf}")(self.h, _ptr(x0), _ptr(U), B, H, C.byref(cost), integ,
                                                         float(dt), _ptr(tb), _ptr(cb), _ptr(gu), _ptr(gx))
        return gu, gx

    def wgrad(self, x, u, lam, Hbar=None):
        """Parameter gradient blob (P,) of sum_p lam_p . f(x_p,u_p) + Hbar_p H(x_p)."""
        x, u, lam = self._a(x, (-1, self.n)), self._a(u, (-1, self.m)), self._a(lam, (-1, self.n))
        hb = None if Hbar is None else self._a(Hbar, (-1,))
        g = np.zeros(self.blob.size, self.dtype)
        getattr(lib(), f"oracle_wgrad_{self.suf}")(self.h, _ptr(x), _ptr(u), _ptr(lam), _ptr(hb), x.shape[0], _ptr(g))
        return g

    def rollout_wgrad(self, x0, U, integrator, dt, traj_bar=None, dx_bar=None):
        """Rollout (no clamp, no cost) + reverse pass with cotangents on the trajectory and on the per-step
        derivatives -> dict(traj, dX, grad_theta (P,), grad_u, grad_x0)."""
        x0 = self._a(x0, (-1, self.n))
        B = x0.shape[0]
        U = self._a(U).reshape(B, -1, self.m)
        H = U.shape[1]
        tb = None if traj_bar is None else self._a(traj_bar, (B, H + 1, self.n))
        db = None if dx_bar is None else self._a(dx_bar, (B, H, self.n))
        traj, dX = np.empty((B, H + 1, self.n), self.dtype), np.empty((B, H, self.n), self.dtype)
        g = np.zeros(self.blob.size, self.dtype)
        gu, gx = np.empty((B, H, self.m), self.dtype), np.empty((B, self.n), self.dtype)
        integ = _capi.INTEGRATORS[integrator] if isinstance(integrator, str) else int(integrator)
        getattr(lib(), f"oracle_rollout_wgrad_{self.suf}")(self.h, _ptr(x0), _ptr(U), B, H, integ, float(dt), _ptr(tb),
                                                           _ptr(db), _ptr(traj), _ptr(dX), _ptr(g), _ptr(gu), _ptr(gx))
        return {"traj": traj, "dX": dX, "grad_theta": g, "grad_u": gu, "grad_x0": gx}

    def adam(self, p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
        """In-place Adam step on arrays of self.dtype."""
        for a in (p, g, m, v):
            assert a.dtype == self.dtype and a.flags.c_contiguous
        getattr(lib(), f"oracle_adam_{self.suf}")(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.size, lr, beta1, beta2, eps, step)


def load_weights(name):
    with np.load(os.path.join(GOLDEN, f"weights_{name}.npz")) as z:
        return {k: z[k] for k in z.files}


def load_wgrad_golden():
    with np.load(os.path.join(GOLDEN, "golden_wgrad.npz")) as z:
        return {k: z[k] for k in z.files}


def load_golden(name):
    with np.load(os.path.join(GOLDEN, f"golden_{name}.npz")) as z:
        return {k: z[k] for k in z.files}


MODELS = ["phnn_cartpole", "canonical_cartpole", "phnn_pendulum", "odefunc_pendulum", "odefunc_cartpole",
          "phnn_cartpole_odd"]  # _odd: hidden widths [96, 80] / [48], zero-padded by the engine to the 128-wide kernels
# 860 epochs of the reference's own training loop on its own dataset (tests/golden/make_trained_cartpole.py).  Kept out
# of MODELS: on these weights random-control rollouts are ill-conditioned (|d cost / d u| up to 3e8) and the REFERENCE's
# own float32 results sit up to 8e-2 (cost) / 3.8 (gradient, of the largest entry) from its float64 ones, so parity is
# judged against that per-rollout noise floor (tests/test_trained_weights.py), not against the fixed tolerances.
TRAINED = "phnn_cartpole_trained"
ROLL_CASES = [(1, 20), (8, 50), (4, 100), (2, 200)]


def trained_rollout_floor(g, key):
    """Per rollout: how far the reference's float32 run is from its float64 run (cost, relative; gradient, of the row's
    largest float64 entry).  The yardstick for any float32 implementation on the trained weights."""
    c64, c32, g64, g32 = g[key + "_cost_f64"], g[key + "_cost_f32"], g[key + "_gu_f64"], g[key + "_gu_f32"]
    gmax = np.abs(g64).max(axis=(1, 2))
    return np.abs(c32 / c64 - 1), np.abs(g32 - g64).max(axis=(1, 2)) / gmax, gmax


def cost_from_golden(g, n=None, m=None, Q=None, x_target=None):
    n = int(g["n"]) if n is None else n
    m = int(g["m"]) if m is None else m
    return _capi.make_cost(n, m, g["Q"] if Q is None else Q, g["R"], g["x_target"] if x_target is None else x_target,
                           float(g["u_min"]), float(g["u_max"]))


def load_m2_golden():
    """golden_m2.npz (two control inputs): -> (arrays, {model name: state_dict})"""
    with np.load(os.path.join(GOLDEN, "golden_m2.npz")) as z:
        g = {k: z[k] for k in z.files}
    w = {}
    for k in list(g):
        if k.startswith("w/"):
            _, name, key = k.split("/", 2)
            w.setdefault(name, {})[key] = g.pop(k)
    return g, w


M2_MODELS = ["phnn_m2_fix", "phnn_m2_gnet", "canonical_m2"]
# models with two, three and four control inputs: (golden file, model name, m)
MULTI_INPUT_MODELS = [("golden_m2.npz", n, 2) for n in M2_MODELS] + [("golden_m34.npz", "phnn_m3_fix", 3),
                                                                      ("golden_m34.npz", "phnn_m4_gnet", 4),
                                                                      ("golden_m34.npz", "canonical_m3", 3)]
# models with other activations than Tanh (tests/golden/make_golden_act.py): name -> activation
ACT_MODELS = {"phnn_silu": "silu", "phnn_relu": "relu", "canonical_silu": "silu", "odefunc_relu": "relu",
              "phnn_elu": "elu", "phnn_gelu": "gelu", "canonical_elu": "elu", "canonical_gelu": "gelu",
              "odefunc_elu": "elu", "odefunc_gelu": "gelu"}


def load_named_golden(fname):
    """golden file with 'w/<model>/<key>' weights -> (arrays, {model name: state_dict})"""
    with np.load(os.path.join(GOLDEN, fname)) as z:
        g = {k: z[k] for k in z.files}
    w = {}
    for k in list(g):
        if k.startswith("w/"):
            _, name, key = k.split("/", 2)
            w.setdefault(name, {})[key] = g.pop(k)
    return g, w


MASS_TYPES = ["constant", "diagonal", "full"]
