/*
 * phnn_mpc.h -- C-ABI of the MI355X-native batched shooting-MPC rollout engine.
 *
 * Drop-in boundary for the hot path of Peilun-Tommy-Li/pHNN-MPC (reference paths below are relative
 * to the reference checkout).  The reference has no FFI of its own: the boundary it exposes is the
 * Python object protocol model(x,u) / euler_step / rollout / compute_cost / cost.backward().  Each
 * entry point here names the reference interface it replaces; INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - plain C, no C++ types or exceptions cross the boundary; every function returns a phnn_status
 *     (0 = ok, negative = error) and records a message readable with phnn_last_error();
 *   - all tensors are contiguous row-major float32; *_dev pointers are device (HBM) addresses owned by
 *     the caller (e.g. torch tensor.data_ptr()); the library owns only the packed weights in a handle;
 *   - every launch is asynchronous on the hipStream_t passed as `void* stream` (NULL = default stream);
 *     no entry point synchronises the device or allocates on the hot path;
 *   - one handle per (model, device); calls on one handle are not re-entrant from several host threads;
 *   - a call leaves the calling thread's current HIP device as it found it (the handle's device is made current
 *     only for the duration of the call).
 */
#ifndef PHNN_MPC_H
#define PHNN_MPC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHNN_MAX_N 8      /* state dimension limit (array bound of the structs below)  */
#define PHNN_MAX_M 4      /* array bound for R in phnn_cost; see PHNN_SUPPORTED_M        */
#define PHNN_SUPPORTED_M 4 /* largest input dimension with gfx950 kernels (m = 2..4: pHNN n = 4 and canonical cart-pole pHNN); phnn_create refuses more */
#define PHNN_MAX_LAYERS 4 /* hidden layers per MLP  */

typedef enum {
  PHNN_OK = 0,
  PHNN_ERR_INVALID_ARG = -1,
  PHNN_ERR_UNSUPPORTED = -2, /* dims / options the HIP kernels are not instantiated for */
  PHNN_ERR_HIP = -3,
  PHNN_ERR_ALLOC = -4
} phnn_status;

/* Dynamics model family (which reference nn.Module the handle restates). */
typedef enum {
  PHNN_MODEL_PHNN = 0,      /* src/pHNN.py:12-100            dx = (J-J^T-R(x)) dH/dx + G u          */
  PHNN_MODEL_CANONICAL = 1, /* src/pHNN_canonical.py:40-273  [q,qdot] -> [q,p] -> back, cart-pole M */
  PHNN_MODEL_ODEFUNC = 2    /* src/baseline_node.py:19-116   dx = MLP_tanh([x,u])                   */
} phnn_model_kind;

typedef enum { PHNN_INTEG_EULER = 0, PHNN_INTEG_RK4 = 1 } phnn_integrator; /* src/integrators.py:13-84 */

/* Mass matrix M(q) of the canonical model (src/pHNN_canonical.py:67-86 selects it from config model.mass_matrix.type):
 *   CARTPOLE  src/mass_matrix.py:239-370  [[a, b cos th],[b cos th, c]], parameters constants to autograd, det + 1e-6
 *   CONSTANT  src/mass_matrix.py:52-57    M = L L^T, L = tril(L_tril) with softplus(diag) + 1e-3; M^-1 = L^-T L^-1
 *   DIAGONAL  src/mass_matrix.py:59-73    M = diag(exp(mlp(q)) + 1e-3)
 *   FULL      src/mass_matrix.py:75-98    M = L(q) L(q)^T, L from mlp(q) (n(n+1)/2 outputs, softplus(diag) + 1e-3),
 *                                         M^-1 = inverse of that 2x2 matrix.   q_dim = 2 (state_dim 4) only. */
typedef enum { PHNN_MASS_CARTPOLE = 0, PHNN_MASS_CONSTANT = 1, PHNN_MASS_DIAGONAL = 2, PHNN_MASS_FULL = 3 } phnn_mass_type;

/* Activation of every MLP in the model (src/NN.py:6-40 takes any nn.Module class -- its default is nn.SiLU;
 * src/pHNN.py:41 resolves it by name; src/baseline_node.py:49-58 offers relu / tanh / elu / gelu).  Tanh is what every
 * shipped config selects and what the fast kernels implement; SiLU, ReLU, ELU (alpha = 1) and GELU (exact, erf form) run
 * on the all-f32 kernels (their activations are unbounded, so the f16 / bf16 split products do not apply; rollouts and
 * VJPs only, no weight-gradient kernels); anything else (OTHER) is refused by phnn_create so that a checkpoint trained with another activation (same
 * keys, same shapes) cannot be run as the wrong network.  One activation per model: all its MLPs must agree. */
typedef enum { PHNN_ACT_TANH = 0, PHNN_ACT_OTHER = 1, PHNN_ACT_SILU = 2, PHNN_ACT_RELU = 3, PHNN_ACT_ELU = 4, PHNN_ACT_GELU = 5 } phnn_activation;

/* How the hidden x hidden products are evaluated.  DEFAULT: f16x2 for 128-wide models, f32 for
 * narrower ones (f16x2 on the 64-wide trained pendulum model is known to exceed the stated tolerance in long
 * rollouts: phnn_create_ex refuses it there unless force_matmul is set). */
typedef enum { PHNN_MATMUL_DEFAULT = 0, PHNN_MATMUL_F32 = 1, PHNN_MATMUL_BF16X3 = 2, PHNN_MATMUL_F16X2 = 3 } phnn_matmul_mode;

/* MLP shape: Linear(in,h[0]) tanh ... Linear(h[depth-1],out); src/NN.py:6-40 with activation nn.Tanh,
 * bias=True, dropout=0, layer_norm=False (the only variant any shipped config selects). */
typedef struct {
  int32_t depth;                   /* number of hidden layers (>=1)      */
  int32_t hidden[PHNN_MAX_LAYERS]; /* hidden sizes                       */
} phnn_mlp_shape;

/* Model description.  The float32 weight blob passed to phnn_create() is the concatenation below, each
 * Linear as weight (out,in) row-major followed by bias (out) -- i.e. the reference state_dict order:
 *   PHNN      : J (n*n) | G_fixed (n*m) if fixed_G | R_net layers | H_net layers | G_net layers if !fixed_G
 *               (src/pHNN.py:22-38; R_net out = n*n, H_net out = 1, G_net out = n*m)
 *   CANONICAL : R_diag_raw (n) | G (n*m) | mass block | H_net layers
 *               (src/pHNN_canonical.py:57-110; J is the fixed canonical one); mass block by mass_type:
 *               CARTPOLE log_a, b, log_c (src/mass_matrix.py:263-268) | CONSTANT L_tril (q_dim*q_dim) |
 *               DIAGONAL / FULL the layers of M_net.mlp (in q_dim, out q_dim / q_dim(q_dim+1)/2)
 *   ODEFUNC   : network layers, in = n+m, out = n (src/baseline_node.py:60-75)
 */
typedef struct {
  int32_t kind;    /* phnn_model_kind */
  int32_t n;       /* state_dim */
  int32_t m;       /* input_dim */
  int32_t fixed_G; /* PHNN only: 1 = G_fixed buffer, 0 = learned G_net */
  phnn_mlp_shape h_net; /* H_net, or the ODEFunc network */
  phnn_mlp_shape r_net; /* PHNN only */
  phnn_mlp_shape g_net; /* PHNN with fixed_G == 0 only */
  int32_t activation;   /* phnn_activation: TANH, SILU or RELU */
  int32_t mass_type;    /* CANONICAL only: phnn_mass_type */
  phnn_mlp_shape m_net; /* CANONICAL with mass_type DIAGONAL / FULL: hidden layers of M_net.mlp */
} phnn_desc;

/* Options of phnn_create_ex; zero-initialise for the defaults. */
typedef struct {
  int32_t matmul_mode;  /* phnn_matmul_mode */
  int32_t force_matmul; /* 1: accept matmul_mode even where it is known to miss the stated tolerance (64-wide + f16x2) */
  int32_t max_waves;    /* waves per workgroup cap, 1..8; 0 = default (8).  4 = one wave per SIMD (diagnostics) */
  int32_t split_tiles;  /* small-batch kernels (four waves share one 16-rollout tile; bitwise the same results):
                         * 0 = automatic (used while the batch has at most 2 tiles per CU), 1 = never, 2 = always
                         * (where the variant has them: 128-wide f16x2 pHNN with fixed G, canonical pHNN) */
  int32_t reserved[4];  /* must be zero */
} phnn_options;

/* Stage cost of both controllers:
 *   cost = sum_{t=0..H} (x_t-x*)^T Q (x_t-x*) + sum_{t<H} u_t^T R u_t
 *        + barrier_weight * sum_{t=0..H} sum_i relu(x_min_i - x_t,i)^2 + relu(x_t,i - x_max_i)^2
 * (src/mpc_controller.py:75-114 with Q = diag, R = scalar*I; src/mpc_controller_canonical.py:91-120 with
 * full Q, R).  Controls are clamped to [u_min,u_max] before use when has_u_bounds, and the returned
 * gradient is w.r.t. the UNclamped controls (torch.clamp backward: pass-through on u_min<=u<=u_max,
 * src/mpc_controller.py:180-181). */
typedef struct {
  float Q[PHNN_MAX_N * PHNN_MAX_N]; /* row-major n x n (leading n*n entries used) */
  float R[PHNN_MAX_M * PHNN_MAX_M]; /* row-major m x m */
  float x_target[PHNN_MAX_N];
  float u_min, u_max;
  int32_t has_u_bounds;
  float x_min[PHNN_MAX_N], x_max[PHNN_MAX_N];
  int32_t has_x_min, has_x_max;
  float barrier_weight; /* reference constant: 1000 */
} phnn_cost;

typedef struct phnn_handle phnn_handle;

/* Build a handle: validates the description, packs the weights into the MFMA fragment order the
 * kernels stage into LDS, and uploads them to `device` once.  Replaces model construction +
 * load_state_dict (src/pHNN.py:13-38, scripts/run_cartpole_mpc.py:27-54). */
int phnn_create(const phnn_desc* desc, const float* weights_host, size_t n_floats, int device,
                phnn_handle** out);
/* Same with explicit options (NULL = defaults).  The library reads no environment variable: the Python host maps
 * PHNN_MATMUL / PHNN_MAX_WAVES to this struct as its own default (phnn_mpc_amd/engine.py). */
int phnn_create_ex(const phnn_desc* desc, const float* weights_host, size_t n_floats, int device,
                   const phnn_options* opt, phnn_handle** out);
/* Replace the weights of an existing handle (same description): re-packs and re-uploads the LDS image on `stream`
 * order.  What load_state_dict / an optimizer step on the reference module amounts to for the engine. */
int phnn_update_weights(phnn_handle* h, const float* weights_host, size_t n_floats, void* stream);
/* The same from a blob that already lives on the handle's device (e.g. the parameters of a module trained there, laid
 * out as for phnn_create): zero-padding to the kernel width and packing run as ONE small kernel in stream order -- no
 * device-to-host copy, host packing or upload, nothing to wait for on the host, and the call may be captured into a HIP
 * graph.  Same image as phnn_update_weights bit for bit, except the constants of CANONICAL models that go through
 * exp / log1p (softplus(R_diag_raw), the mass-matrix constants), where host and device math libraries may differ in the
 * last bit.  Kernels launched later on `stream` see the new weights; work on other streams must be ordered by the caller. */
int phnn_update_weights_dev(phnn_handle* h, const float* weights_dev, size_t n_floats, void* stream);
int phnn_destroy(phnn_handle* h);
/* Message of the last failing call on this handle (or of the last failing phnn_create if h == NULL). */
const char* phnn_last_error(const phnn_handle* h);
/* Number of float32 the weight blob for `desc` must hold; 0 if the description is invalid. */
size_t phnn_weight_count(const phnn_desc* desc);

/* model(x,u) -> (dx, H): src/pHNN.py:52-100, src/pHNN_canonical.py:172-273, src/baseline_node.py:88-116.
 * x_dev (B,n), u_dev (B,m) -> dx_dev (B,n), H_dev (B) (H_dev may be NULL; ODEFUNC writes 0). */
int phnn_model_forward(phnn_handle* h, const float* x_dev, const float* u_dev, int64_t B, float* dx_dev,
                       float* H_dev, void* stream);

/* Vector-Jacobian product of the dynamics, what autograd does for one model call inside
 * cost.backward(): xbar = (df/dx)^T lam, ubar = (df/du)^T lam.  lam_dev (B,n) -> xbar_dev (B,n),
 * ubar_dev (B,m). */
int phnn_model_vjp(phnn_handle* h, const float* x_dev, const float* u_dev, const float* lam_dev, int64_t B,
                   float* xbar_dev, float* ubar_dev, void* stream);

/* K1 -- fused forward march: clamp, dynamics, integrator step and stage cost over the whole horizon.
 * Replaces rollout_dynamics + compute_cost (src/mpc_controller.py:75-141), rollout + compute_cost
 * (src/mpc_controller_canonical.py:91-161) and rollout_trajectory_differentiable (src/integrators.py:192-258).
 *   x0_dev (B,n), u_dev (B,H,m) -> cost_dev (B), traj_dev (B,H+1,n) (may be NULL when neither the
 *   trajectory nor a later phnn_rollout_grad is wanted). */
int phnn_rollout_fwd(phnn_handle* h, const float* x0_dev, const float* u_dev, int64_t B, int32_t H,
                     const phnn_cost* cost, int32_t integrator, float dt, float* cost_dev, float* traj_dev,
                     void* workspace_dev, void* stream);

/* Optional K1 -> K2 workspace.  With workspace_dev != NULL (at least phnn_workspace_bytes() bytes) K1 also streams
 * the H_net / network activations of every dynamics evaluation to it and K2, given the same pointer, reads them back
 * instead of re-evaluating the forward pass: about half of K2's matrix work for 1.1 KB per rollout-step of extra HBM
 * traffic each way (cart-pole, Euler).  RK4 keeps four stage slots per step (tape + stage state: 4.4 KB per
 * rollout-step), so that K2 runs no forward evaluation at all; pass NULL to trade that memory for recomputation.
 * The workspace passed to phnn_rollout_grad / phnn_rollout_vjp must have been filled by phnn_rollout_fwd with the
 * same (x0, u, B, H, cost, integrator, dt). */
size_t phnn_workspace_bytes(const phnn_handle* h, int64_t B, int32_t H, int32_t integrator);

/* K2 -- adjoint march: replaces cost.backward() (src/mpc_controller.py:192,
 * src/mpc_controller_canonical.py:205).  Reads the states K1 wrote to traj_dev, re-evaluates the
 * dynamics at each of them, and marches the costate backwards (explicit Hessian-vector product of
 * H_net).  -> grad_u_dev (B,H,m) = d cost_b / d u_b (unclamped), grad_x0_dev (B,n) or NULL. */
int phnn_rollout_grad(phnn_handle* h, const float* x0_dev, const float* u_dev, int64_t B, int32_t H,
                      const phnn_cost* cost, int32_t integrator, float dt, const float* traj_dev,
                      const void* workspace_dev, float* grad_u_dev, float* grad_x0_dev, void* stream);

/* General reverse pass of the rollout, what autograd does when a loss is built on BOTH outputs of
 * rollout_trajectory_differentiable + compute_cost (src/integrators.py:192-258): given cotangents
 * traj_bar_dev (B,H+1,n) on the trajectory (may be NULL = 0) and cost_bar_dev (B) on the cost (may be NULL = 1),
 * returns grad_u_dev (B,H,m) and grad_x0_dev (B,n) (may be NULL).  phnn_rollout_grad is the special case
 * traj_bar = 0, cost_bar = 1. */
int phnn_rollout_vjp(phnn_handle* h, const float* x0_dev, const float* u_dev, int64_t B, int32_t H,
                     const phnn_cost* cost, int32_t integrator, float dt, const float* traj_dev,
                     const void* workspace_dev, const float* traj_bar_dev, const float* cost_bar_dev,
                     float* grad_u_dev, float* grad_x0_dev, void* stream);

/* ---- training side: rollouts with gradients w.r.t. the MODEL PARAMETERS ----------------------
 * What loss.backward() does in the reference's training loops (scripts/train_cartpole_phnn.py:112-178,
 * scripts/train_cartpole_phnn_canonical.py:83-196, main.py:93-148): an Euler (or RK4) rollout from x_batch[:,0] with
 * the dataset's controls, a loss built on the predicted states X_pred and/or the per-step derivatives dX_pred, and the
 * gradient of that loss w.r.t. every parameter.  The loss itself stays with the caller (torch): these entry points take
 * its cotangents on the two outputs and return the parameter gradient as a blob laid out exactly like the weight blob
 * of phnn_create (buffers of the reference modules -- G_fixed, the canonical G -- and the CartPoleMassMatrix
 * parameters, which are constants to autograd through .item(), src/mass_matrix.py:299-301, stay zero).
 * Available for PHNN and CANONICAL handles (PHNN_ERR_UNSUPPORTED for ODEFUNC). */

/* Forward of a training rollout: no clamp, no cost.  -> traj_dev (B,H+1,n) = X_pred, dx_dev (B,H,n) = dX_pred
 * (f(x_t,u_t) at the first stage of each step; may be NULL). */
int phnn_rollout_trajectory(phnn_handle* h, const float* x0_dev, const float* u_dev, int64_t B, int32_t H,
                            int32_t integrator, float dt, float* traj_dev, float* dx_dev, void* stream);

/* The same, and with wgrad_workspace_dev != NULL (phnn_wgrad_workspace_bytes() bytes, the buffer later handed to
 * phnn_rollout_wgrad) it also keeps the tapes of every dynamics evaluation in that workspace, so that
 * phnn_rollout_wgrad(..., flags | PHNN_WGRAD_TAPES) neither re-evaluates the forward pass nor copies the tapes into
 * its records (training pass 6.6 -> 5.x ms at B = 65536, H = 50). */
int phnn_rollout_trajectory_ws(phnn_handle* h, const float* x0_dev, const float* u_dev, int64_t B, int32_t H,
                               int32_t integrator, float dt, float* traj_dev, float* dx_dev, void* wgrad_workspace_dev,
                               void* stream);

/* Bytes of workspace phnn_rollout_wgrad (H >= 1) / phnn_model_wgrad (H = 0, B = number of points) need. */
size_t phnn_wgrad_workspace_bytes(const phnn_handle* h, int64_t B, int32_t H, int32_t integrator);

/* flags of phnn_rollout_wgrad / phnn_model_wgrad */
#define PHNN_WGRAD_ACCUMULATE 1 /* add to grad_theta_dev instead of overwriting it */
#define PHNN_WGRAD_TAPES 2      /* the workspace holds the tapes phnn_rollout_trajectory_ws wrote for the SAME
                                   (x0, u, B, H, integrator, dt) and weights; anything else gives wrong gradients */

/* Layout of the records phnn_rollout_wgrad / phnn_model_wgrad leave at the start of the workspace: record r (one per
 * 16-point tile and evaluation: tile-major, then step, then RK4 stage) starts at float r * record_floats; its
 * per-point block starts small_offset floats in and holds small_stride floats for each of the tile's 16 points.  For
 * CANONICAL handles with a MassMatrixNetwork (mass_type != PHNN_MASS_CARTPOLE) floats 20..23 of a point's block are the
 * cotangent of the 2 x 2 matrix M(q) (row-major) of that evaluation and floats 24, 25 its q: the gradient of the mass
 * network's parameters -- which the kernels leave at zero in grad_theta -- is one autograd pass of the caller's
 * MassMatrixNetwork over those points (phnn_mpc_amd/models.py does exactly that); points beyond B carry zeros. */
int phnn_wgrad_record_info(const phnn_handle* h, int32_t* record_floats, int32_t* small_offset, int32_t* small_stride);

/* Reverse pass of the training rollout.  traj_dev: the states phnn_rollout_trajectory wrote; traj_bar_dev (B,H+1,n) and
 * dx_bar_dev (B,H,n): cotangents of the loss on X_pred and dX_pred (either may be NULL = 0).
 * -> grad_theta_dev (phnn_weight_count floats; overwritten, or added to with PHNN_WGRAD_ACCUMULATE), grad_u_dev (B,H,m)
 * and grad_x0_dev (B,n) (both may be NULL).  Two kernels: the adjoint march, which also streams one record per
 * (16-rollout tile, step, stage) to the workspace, and a reduction of the records into the gradient (GEMMs over the
 * evaluation points; fixed summation order, bitwise reproducible). */
int phnn_rollout_wgrad(phnn_handle* h, const float* x0_dev, const float* u_dev, int64_t B, int32_t H, int32_t integrator,
                       float dt, const float* traj_dev, const float* traj_bar_dev, const float* dx_bar_dev,
                       void* workspace_dev, float* grad_theta_dev, int32_t flags, float* grad_u_dev,
                       float* grad_x0_dev, void* stream);

/* Single evaluations: gradient of sum_p lam_p . f(x_p,u_p) + Hbar_p H(x_p) w.r.t. the parameters (what backward() does
 * for model(x,u) calls outside a rollout, e.g. the energy anchor H(0)^2 of scripts/train_cartpole_phnn.py:166-170),
 * plus xbar_dev (N,n) and ubar_dev (N,m) as phnn_model_vjp (with Hbar dH/dx added).  Hbar_dev (N) may be NULL. */
int phnn_model_wgrad(phnn_handle* h, const float* x_dev, const float* u_dev, const float* lam_dev, const float* Hbar_dev,
                     int64_t N, void* workspace_dev, float* grad_theta_dev, int32_t accumulate, float* xbar_dev,
                     float* ubar_dev, void* stream);

/* K3 -- Adam step on the controls, arithmetic order of torch.optim.Adam (single-tensor, defaults:
 * no weight decay / amsgrad) as used by src/mpc_controller.py:168,200 and
 * src/mpc_controller_canonical.py:186,206.  `step` is the 1-based step count after the increment.
 * Optionally fuses the best-iterate tracking of src/mpc_controller_canonical.py:208-214: when
 * best_cost_dev != NULL, rollouts whose cost_dev[b] < best_cost_dev[b] copy clamp(u_b) (the pre-step
 * iterate that produced cost_dev) into best_u_dev and update best_cost_dev.  count = B*H*m, per = H*m. */
int phnn_adam_step(phnn_handle* h, float* u_dev, const float* grad_dev, float* exp_avg_dev,
                   float* exp_avg_sq_dev, int64_t count, float lr, float beta1, float beta2, float eps,
                   int32_t step, const float* cost_dev, float* best_cost_dev, float* best_u_dev, int64_t per,
                   float u_min, float u_max, int32_t has_u_bounds, void* stream);

/* The whole shooting solve behind MPCController.compute_control (src/mpc_controller.py:164-209) and
 * MPCControllerCanonical.optimize_control (src/mpc_controller_canonical.py:163-228) for B independent problems:
 * `iters` times { cost and gradient of the current iterate (K1, K2); Adam step (K3) }, a fresh optimizer state per call.
 *   u_dev (B,H,m): in = initial iterate (zeros for a cold start, the shifted previous solution for a warm start),
 *                  out = the LAST iterate, unclamped (compute_control returns clamp(u[0]) of it);
 *   track_best: best_u_dev (B,H,m) / best_cost_dev (B) receive the clamped iterate of lowest cost, the cost of iterate k
 *               being measured before step k is applied, strict '<' (what optimize_control returns);
 *   costs_dev (iters,B) or NULL: the cost history (info['costs'] of the canonical controller);
 *   exp_avg_dev, exp_avg_sq_dev, grad_dev (B,H,m), cost_dev (B), traj_dev (B,H+1,n): caller-owned scratch, overwritten;
 *   workspace_dev: phnn_workspace_bytes() bytes (K1 -> K2 tape) or NULL (the adjoint recomputes).
 * The call enqueues the 3 x iters launches on `stream` (stream-ordered, no host synchronisation; small batches run on the
 * split-tile kernels).  A fused one-launch form was measured and dropped: back-to-back launches are already pipelined, a
 * single-plant solve is the serial chain of its time steps. */
typedef struct {
  int32_t iters;
  float lr, beta1, beta2, eps; /* torch.optim.Adam: lr, betas (0.9, 0.999), eps 1e-8 */
  int32_t track_best;
} phnn_solve_options;
int phnn_solve(phnn_handle* h, const float* x0_dev, float* u_dev, int64_t B, int32_t H, const phnn_cost* cost,
               int32_t integrator, float dt, const phnn_solve_options* opt, float* exp_avg_dev, float* exp_avg_sq_dev,
               float* grad_dev, float* cost_dev, float* traj_dev, void* workspace_dev, float* costs_dev,
               float* best_cost_dev, float* best_u_dev, void* stream);

/* ---- the plant on the other side of the path ------------------------------------------
 * Ground-truth cart-pole of src/cartpole_simulator.py:63-112: float64, explicit Euler, the standard cart-pole
 * equations in the reference's operation order; termination |x| > x_limit or |theta| > theta_limit.  Defaults of
 * the reference: gravity 9.8, masscart 1.0, masspole 0.1, length 0.5 (half-length), dt 0.02, limits 10.0 / 0.5. */
typedef struct {
  double gravity, masscart, masspole, length, dt, x_limit, theta_limit;
} phnn_plant;

/* One plant step for B plants, everything resident on the device (no host round trip in a closed loop):
 *   state_dev (B,4) float64, updated in place;  action: action_dev[b * action_stride] (float32 force; the first
 *   control of rollout b's sequence when action_stride = H*m), clamped to [u_min,u_max] first when has_u_bounds
 *   (src/mpc_controller.py:203-209 returns clamp(u_seq[0])).
 * Optional outputs (NULL = skip): state_f32_dev (B,4) = float32 of the new state (what the next solve consumes:
 * torch.tensor(state, dtype=float32) in scripts/run_cartpole_mpc.py:129); done_step_dev (B) int32, set to the step
 * index the first time a plant terminates (initialise to -1); log_states_dev (T+1,B,4) float64 and
 * log_controls_dev (T,B) float32 receive row step+1 / row step.  The step index is *step_dev when step_dev != NULL
 * (a device counter, so that a captured HIP graph can be replayed), else step_host. */
int phnn_plant_step(phnn_handle* h, const phnn_plant* plant, double* state_dev, const float* action_dev,
                    int64_t action_stride, int64_t B, int32_t has_u_bounds, float u_min, float u_max,
                    float* state_f32_dev, int32_t* done_step_dev, const int32_t* step_dev, int32_t step_host,
                    double* log_states_dev, float* log_controls_dev, void* stream);

/* Warm start of the next solve, src/mpc_controller_canonical.py:252-255: dst[b,t] = src[b,t+1] for t < H-1,
 * dst[b,H-1] = 0 (u (B,H,m), shift by one step).  Also advances *step_dev by one when step_dev != NULL. */
int phnn_shift_controls(phnn_handle* h, const float* src_dev, float* dst_dev, int64_t B, int32_t H, int32_t m,
                        int32_t* step_dev, void* stream);

/* Introspection for tests: copies the packed weight image the kernels stage into LDS (phnn_kernels.hip.h layouts; a
 * device-resident private format) to image_host after the work queued on `stream`, and waits for the copy.
 * *image_floats (may be NULL) receives its size; image_host == NULL only queries the size. */
int phnn_read_image(phnn_handle* h, float* image_host, size_t n_floats, size_t* image_floats, void* stream);

/* Introspection for benches/tests: name of the kernel variant selected for this handle, rollouts per
 * workgroup, LDS bytes staged per workgroup. */
int phnn_kernel_info(const phnn_handle* h, int32_t integrator, int32_t* rollouts_per_wg,
                     int32_t* lds_bytes, int32_t* n_workgroups_for_B, int64_t B);

/* Name of the kernel variant serving this handle, e.g. "phnn<n=4,hid=128,fixedG,f16x2>".  The hidden x hidden
 * products run as all-f32 MFMAs ("f32"), as an exact 3-way bf16 split ("bf16x3") or 2-way f16 split ("f16x2",
 * default for 128-wide models) on the matrix pipe; phnn_options.matmul_mode selects one at phnn_create_ex time. */
const char* phnn_variant_name(const phnn_handle* h);

/* Library version (major*10000 + minor*100 + patch). */
int phnn_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PHNN_MPC_H */
