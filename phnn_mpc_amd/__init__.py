"""phnn_mpc_amd -- MI355X-native batched shooting-MPC rollout engine (drop-in for the pHNN-MPC hot path).

Importing the package does not load the HIP library; the engine does, and raises if it is missing.
"""
__version__ = "0.1.0"
