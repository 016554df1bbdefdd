"""Shared inputs of the L1-chain tests (CPU: tests/test_oracle_chain.py, GPU: tests/test_gpu_chain.py); bench.py's w1full workload
draws the same stream from eorb_slam_amd.synth."""
from eorb_slam_amd import synth

CAM = synth.EVETHZ_PINHOLE
stream = synth.l1_stream
mci_poses = synth.l1_mci_poses
