"""MI355X-native host package for the two-tower diffusion LCM generation path.

Keeps the public names of the reference's `lcm` package for this path
(`lcm.inference.two_tower_diffusion_lcm`, `lcm.models.two_tower_diffusion_lcm`,
`lcm.datasets.batch.EmbeddingsBatch`, `lcm.utils.card_utils.load_model_from_card`) and runs all
arithmetic in hand-written gfx950 kernels behind the C ABI of `liblcm_hip.so`
(see include/lcmhip.h).  PyTorch only owns device memory and streams.
"""

__version__ = "0.1.0"
