"""rlcontrol_amd -- MI355X-native replay-sampling + actor-critic-update hot path of RLControl.

Layout
  csrc/            hand-written HIP kernels (gfx950) + the C ABI of librlcontrol_hip.so
  hip_ddpg.py      typed Python face of one rlc_ddpg handle (a population of agents on one GPU)
  agents/, utils/, experiment.py, environments/   host-side mirror of the reference's interfaces
The package has no CPU compute path: without librlcontrol_hip.so and an MI355X it raises.
"""
__version__ = "0.1.0"
