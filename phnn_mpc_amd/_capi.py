"""ctypes view of include/phnn_mpc.h: the structs, and the loader of the HIP shared library.

The product path has no CPU fallback: if libphnn_mpc.so is missing or a symbol is absent the import of the
engine raises, loudly.
"""
import ctypes as C
import os

PHNN_MAX_N = 8
PHNN_MAX_M = 4
PHNN_MAX_LAYERS = 4

MODEL_PHNN, MODEL_CANONICAL, MODEL_ODEFUNC = 0, 1, 2
INTEG_EULER, INTEG_RK4 = 0, 1
INTEGRATORS = {"euler": INTEG_EULER, "rk4": INTEG_RK4}

ACT_TANH, ACT_OTHER, ACT_SILU, ACT_RELU, ACT_ELU, ACT_GELU = 0, 1, 2, 3, 4, 5
ACTIVATIONS = {"tanh": ACT_TANH, "silu": ACT_SILU, "relu": ACT_RELU, "elu": ACT_ELU, "gelu": ACT_GELU}  # activations with kernels (others: ACT_OTHER, refused)
MASS_CARTPOLE, MASS_CONSTANT, MASS_DIAGONAL, MASS_FULL = 0, 1, 2, 3
WGRAD_ACCUMULATE, WGRAD_TAPES = 1, 2  # flags of phnn_rollout_wgrad / phnn_model_wgrad (include/phnn_mpc.h)
MATMUL_MODES = {"default": 0, "f32": 1, "bf16x3": 2, "f16x2": 3}
SPLIT_MODES = {"auto": 0, "never": 1, "always": 2}

_HERE = os.path.dirname(os.path.abspath(__file__))
# PHNN_LIB_PATH: load another build of the library (A/B comparisons, tools/ab_bench.sh) without touching the product file
LIB_PATH = os.environ.get("PHNN_LIB_PATH") or os.path.join(_HERE, "csrc", "libphnn_mpc.so")

# every symbol include/phnn_mpc.h declares
EXPORTED = [
    "phnn_create", "phnn_create_ex", "phnn_update_weights", "phnn_update_weights_dev", "phnn_read_image", "phnn_destroy", "phnn_last_error", "phnn_weight_count", "phnn_model_forward",
    "phnn_model_vjp", "phnn_rollout_fwd", "phnn_workspace_bytes", "phnn_rollout_grad", "phnn_rollout_vjp",
    "phnn_rollout_trajectory", "phnn_rollout_trajectory_ws", "phnn_wgrad_workspace_bytes", "phnn_wgrad_record_info", "phnn_rollout_wgrad", "phnn_model_wgrad",
    "phnn_adam_step", "phnn_solve", "phnn_plant_step", "phnn_shift_controls", "phnn_kernel_info", "phnn_variant_name",
    "phnn_version",
]


class MlpShape(C.Structure):
    _fields_ = [("depth", C.c_int32), ("hidden", C.c_int32 * PHNN_MAX_LAYERS)]

    @classmethod
    def of(cls, hidden):
        hidden = list(hidden)
        if len(hidden) > PHNN_MAX_LAYERS:
            raise ValueError(f"at most {PHNN_MAX_LAYERS} hidden layers are supported, got {len(hidden)}")
        s = cls()
        s.depth = len(hidden)
        for i, h in enumerate(hidden):
            s.hidden[i] = int(h)
        return s


class Desc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n", C.c_int32), ("m", C.c_int32), ("fixed_G", C.c_int32),
                ("h_net", MlpShape), ("r_net", MlpShape), ("g_net", MlpShape), ("activation", C.c_int32),
                ("mass_type", C.c_int32), ("m_net", MlpShape)]


class SolveOptions(C.Structure):
    """phnn_solve_options"""
    _fields_ = [("iters", C.c_int32), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("track_best", C.c_int32)]


class Plant(C.Structure):
    """phnn_plant: the reference's cart-pole constants (src/cartpole_simulator.py:26-36, 107-110)."""
    _fields_ = [("gravity", C.c_double), ("masscart", C.c_double), ("masspole", C.c_double), ("length", C.c_double),
                ("dt", C.c_double), ("x_limit", C.c_double), ("theta_limit", C.c_double)]

    @classmethod
    def default(cls, dt=0.02):
        return cls(9.8, 1.0, 0.1, 0.5, float(dt), 10.0, 0.5)


class Options(C.Structure):
    _fields_ = [("matmul_mode", C.c_int32), ("force_matmul", C.c_int32), ("max_waves", C.c_int32),
                ("split_tiles", C.c_int32), ("reserved", C.c_int32 * 4)]


class Cost(C.Structure):
    _fields_ = [("Q", C.c_float * (PHNN_MAX_N * PHNN_MAX_N)), ("R", C.c_float * (PHNN_MAX_M * PHNN_MAX_M)),
                ("x_target", C.c_float * PHNN_MAX_N), ("u_min", C.c_float), ("u_max", C.c_float),
                ("has_u_bounds", C.c_int32), ("x_min", C.c_float * PHNN_MAX_N), ("x_max", C.c_float * PHNN_MAX_N),
                ("has_x_min", C.c_int32), ("has_x_max", C.c_int32), ("barrier_weight", C.c_float)]


def make_cost(n, m, Q, R, x_target=None, u_min=None, u_max=None, x_min=None, x_max=None, barrier_weight=1000.0):
    """Fill a phnn_cost.  Q: (n,n) matrix or length-n diagonal; R: (m,m) matrix, length-m diagonal or scalar."""
    import numpy as np

    c = Cost()
    Q = np.asarray(Q, dtype=np.float64)
    if Q.ndim == 1:
        Q = np.diag(Q)
    if Q.shape != (n, n):
        raise ValueError(f"Q must be ({n},{n}) or ({n},), got {Q.shape}")
    R = np.asarray(R, dtype=np.float64)
    if R.ndim == 0:
        R = np.eye(m) * float(R)
    elif R.ndim == 1:
        R = np.diag(R)
    if R.shape != (m, m):
        raise ValueError(f"R must be ({m},{m}), ({m},) or scalar, got {R.shape}")
    for i in range(n):
        for j in range(n):
            c.Q[i * n + j] = Q[i, j]
    for i in range(m):
        for j in range(m):
            c.R[i * m + j] = R[i, j]
    xt = np.zeros(n) if x_target is None else np.asarray(x_target, dtype=np.float64).reshape(n)
    for i in range(n):
        c.x_target[i] = xt[i]
    if (u_min is None) != (u_max is None):
        # the reference only clamps when both bounds are given (src/mpc_controller.py:180-183)
        u_min = u_max = None
    c.has_u_bounds = int(u_min is not None)
    c.u_min = float(u_min) if u_min is not None else 0.0
    c.u_max = float(u_max) if u_max is not None else 0.0
    c.has_x_min = int(x_min is not None)
    c.has_x_max = int(x_max is not None)
    if x_min is not None:
        for i, v in enumerate(np.asarray(x_min, dtype=np.float64).reshape(n)):
            c.x_min[i] = v
    if x_max is not None:
        for i, v in enumerate(np.asarray(x_max, dtype=np.float64).reshape(n)):
            c.x_max[i] = v
    c.barrier_weight = float(barrier_weight)
    return c


_lib = None


def load_library():
    """Load csrc/libphnn_mpc.so and declare the signatures of include/phnn_mpc.h.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C phnn_mpc_amd/csrc`). "
            "There is no CPU fallback for the rollout engine.")
    # torch first: it brings its own HIP runtime (same soname as the system one the library is linked against);
    # the runtime that is loaded first serves both, and only torch's matches the rest of torch's ROCm libraries.
    # Loading this library before torch leaves the process with a mixed set and no visible device.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    vp, f32p, i64, i32 = C.c_void_p, C.c_void_p, C.c_int64, C.c_int32
    lib.phnn_create.argtypes = [C.POINTER(Desc), C.POINTER(C.c_float), C.c_size_t, C.c_int, C.POINTER(vp)]
    lib.phnn_create.restype = C.c_int
    lib.phnn_create_ex.argtypes = [C.POINTER(Desc), C.POINTER(C.c_float), C.c_size_t, C.c_int, C.POINTER(Options),
                                   C.POINTER(vp)]
    lib.phnn_create_ex.restype = C.c_int
    lib.phnn_update_weights.argtypes = [vp, C.POINTER(C.c_float), C.c_size_t, vp]
    lib.phnn_update_weights.restype = C.c_int
    lib.phnn_update_weights_dev.argtypes = [vp, f32p, C.c_size_t, vp]
    lib.phnn_update_weights_dev.restype = C.c_int
    lib.phnn_read_image.argtypes = [vp, C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_size_t), vp]
    lib.phnn_read_image.restype = C.c_int
    lib.phnn_destroy.argtypes = [vp]
    lib.phnn_destroy.restype = C.c_int
    lib.phnn_last_error.argtypes = [vp]
    lib.phnn_last_error.restype = C.c_char_p
    lib.phnn_weight_count.argtypes = [C.POINTER(Desc)]
    lib.phnn_weight_count.restype = C.c_size_t
    lib.phnn_model_forward.argtypes = [vp, f32p, f32p, i64, f32p, f32p, vp]
    lib.phnn_model_forward.restype = C.c_int
    lib.phnn_model_vjp.argtypes = [vp, f32p, f32p, f32p, i64, f32p, f32p, vp]
    lib.phnn_model_vjp.restype = C.c_int
    lib.phnn_rollout_fwd.argtypes = [vp, f32p, f32p, i64, i32, C.POINTER(Cost), i32, C.c_float, f32p, f32p, vp, vp]
    lib.phnn_rollout_fwd.restype = C.c_int
    lib.phnn_workspace_bytes.argtypes = [vp, i64, i32, i32]
    lib.phnn_workspace_bytes.restype = C.c_size_t
    lib.phnn_rollout_grad.argtypes = [vp, f32p, f32p, i64, i32, C.POINTER(Cost), i32, C.c_float, f32p, vp, f32p, f32p,
                                      vp]
    lib.phnn_rollout_grad.restype = C.c_int
    lib.phnn_rollout_vjp.argtypes = [vp, f32p, f32p, i64, i32, C.POINTER(Cost), i32, C.c_float, f32p, vp, f32p, f32p,
                                     f32p, f32p, vp]
    lib.phnn_rollout_vjp.restype = C.c_int
    lib.phnn_rollout_trajectory.argtypes = [vp, f32p, f32p, i64, i32, i32, C.c_float, f32p, f32p, vp]
    lib.phnn_rollout_trajectory.restype = C.c_int
    lib.phnn_rollout_trajectory_ws.argtypes = [vp, f32p, f32p, i64, i32, i32, C.c_float, f32p, f32p, vp, vp]
    lib.phnn_rollout_trajectory_ws.restype = C.c_int
    lib.phnn_wgrad_record_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    lib.phnn_wgrad_record_info.restype = C.c_int
    lib.phnn_wgrad_workspace_bytes.argtypes = [vp, i64, i32, i32]
    lib.phnn_wgrad_workspace_bytes.restype = C.c_size_t
    lib.phnn_rollout_wgrad.argtypes = [vp, f32p, f32p, i64, i32, i32, C.c_float, f32p, f32p, f32p, vp, f32p, i32, f32p,
                                       f32p, vp]
    lib.phnn_rollout_wgrad.restype = C.c_int
    lib.phnn_model_wgrad.argtypes = [vp, f32p, f32p, f32p, f32p, i64, vp, f32p, i32, f32p, f32p, vp]
    lib.phnn_model_wgrad.restype = C.c_int
    lib.phnn_adam_step.argtypes = [vp, f32p, f32p, f32p, f32p, i64, C.c_float, C.c_float, C.c_float, C.c_float, i32,
                                   f32p, f32p, f32p, i64, C.c_float, C.c_float, i32, vp]
    lib.phnn_adam_step.restype = C.c_int
    lib.phnn_solve.argtypes = [vp, f32p, f32p, i64, i32, C.POINTER(Cost), i32, C.c_float, C.POINTER(SolveOptions), f32p, f32p,
                               f32p, f32p, f32p, vp, f32p, f32p, f32p, vp]
    lib.phnn_solve.restype = C.c_int
    lib.phnn_plant_step.argtypes = [vp, C.POINTER(Plant), vp, f32p, i64, i64, i32, C.c_float, C.c_float, f32p, vp, vp, i32,
                                    vp, f32p, vp]
    lib.phnn_plant_step.restype = C.c_int
    lib.phnn_shift_controls.argtypes = [vp, f32p, f32p, i64, i32, i32, vp, vp]
    lib.phnn_shift_controls.restype = C.c_int
    lib.phnn_kernel_info.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), i64]
    lib.phnn_kernel_info.restype = C.c_int
    lib.phnn_variant_name.argtypes = [vp]
    lib.phnn_variant_name.restype = C.c_char_p
    lib.phnn_version.argtypes = []
    lib.phnn_version.restype = C.c_int
    _lib = lib
    return lib
