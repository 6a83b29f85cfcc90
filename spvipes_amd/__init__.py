"""spvipes_amd -- MI355X-native hot path of spVIPES (per-minibatch shared/private PoE VAE step).

The compute path is hand-written HIP for gfx950 behind a C ABI (include/spvipes_hip.h); PyTorch
supplies device memory, streams, autograd plumbing and torch.distributed only.
"""
__version__ = "0.1.0"
