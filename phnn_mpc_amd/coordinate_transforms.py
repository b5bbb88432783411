"""Drop-in for src/coordinate_transforms.py:20-237: conversions between the kinematic state y = [q, qdot] and the
canonical state z = [q, p] through a mass-matrix module (M_net(q) -> (B,d,d), M_net.inverse(q)).

These are host-side utilities with the reference's names and signatures (diagnostics, velocity-reconstruction
losses).  The rollout kernels evaluate the same transforms in-kernel for the cart-pole mass matrix
(pHNN_Canonical.forward, src/pHNN_canonical.py:172-273) and do not call into this module.
"""
import torch


def velocity_to_momentum(q, q_dot, M_net):
    """p = M(q) qdot"""
    return torch.bmm(M_net(q), q_dot.unsqueeze(-1)).squeeze(-1)


def momentum_to_velocity(q, p, M_net):
    """qdot = M(q)^-1 p"""
    return torch.bmm(M_net.inverse(q), p.unsqueeze(-1)).squeeze(-1)


def split_state(state):
    """(B, 2d) -> position (B,d), velocity-or-momentum (B,d)"""
    d = state.shape[1] // 2
    return state[:, :d], state[:, d:]


def kinematic_to_canonical(y, M_net):
    """[q, qdot] -> [q, p]"""
    q, q_dot = split_state(y)
    return torch.cat([q, velocity_to_momentum(q, q_dot, M_net)], dim=1)


def canonical_to_kinematic(z, M_net):
    """[q, p] -> [q, qdot]"""
    q, p = split_state(z)
    return torch.cat([q, momentum_to_velocity(q, p, M_net)], dim=1)


def combine_state(q, v):
    """position (B,d), velocity-or-momentum (B,d) -> (B, 2d)   (src/coordinate_transforms.py:133-144)"""
    return torch.cat([q, v], dim=1)


def batch_matrix_vector_product(A, v):
    """(B,n,n), (B,n) -> A v (B,n)   (src/coordinate_transforms.py:147-161)"""
    return torch.bmm(A, v.unsqueeze(-1)).squeeze(-1)


def compute_kinetic_energy(q, p, M_net):
    """T = p^T M(q)^-1 p / 2, (B,)   (src/coordinate_transforms.py:164-183)"""
    return 0.5 * torch.sum(p * momentum_to_velocity(q, p, M_net), dim=1)


def verify_coordinate_transform(y, M_net, tol=1e-5):
    """Round trip y -> z -> y: (max |y - y'| < tol, max |y - y'|)   (src/coordinate_transforms.py:186-212)"""
    back = canonical_to_kinematic(kinematic_to_canonical(y, M_net), M_net)
    worst = torch.max(torch.abs(y - back)).item()
    return worst < tol, worst


def compute_velocity_reconstruction_error(q, q_dot_true, p, M_net):
    """|| M(q)^-1 p - qdot_true ||^2 per sample, (B,): a term of the canonical training loss
    (src/coordinate_transforms.py:215-237)"""
    return torch.sum((momentum_to_velocity(q, p, M_net) - q_dot_true) ** 2, dim=1)
