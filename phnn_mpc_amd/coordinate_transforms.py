"""Drop-in for src/coordinate_transforms.py:20-130: conversions between the kinematic state y = [q, qdot] and the
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
