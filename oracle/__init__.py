"""CPU oracle for the two-tower diffusion LCM generation path.

TEST INFRASTRUCTURE ONLY.  This package is a plain PyTorch-CPU (fp32) restatement of the
reference algorithm (viasky657/Chimera-LLM, `lcm/inference/two_tower_diffusion_lcm`,
`lcm/models/two_tower_diffusion_lcm`, `lcm/nn/*`).  It exists to check the HIP path and to
provide the CPU baseline timing.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it; the product package (`lcm`) never does.

PARITY UNPINNED: the reference cannot be imported here (its third-party runtime, fairseq2
0.3.0rc1, is absent and un-fetchable) and its test-suite holds no numeric golden vectors for
this path.  The oracle is therefore pinned only by (a) the reference's own *invariant* tests
restated for this path (KV cache on/off, batched vs. single, determinism), (b) analytic
identities of the algorithm, and (c) the one data fixture the reference ships
(`lcm/cards/mock_data/dummy_normalizer.pt`, copied as data to `tests/golden/`).

Each function cites the reference file:line it follows (paths relative to the reference root).
"""

from .archs import ArchConfig, get_arch, ARCHS, BaseArchConfig, get_base_arch, BASE_ARCHS  # noqa: F401
from .weights import init_state_dict, load_dummy_normalizer  # noqa: F401
from .scheduler import DDIMSchedule  # noqa: F401
from .model import OracleTwoTower, OracleBaseLCM  # noqa: F401
from .bf16_emu import OracleTwoTowerBf16, OracleBaseLCMBf16, r16, summation_order  # noqa: F401
from .generator import OracleGenerator, OracleScorer, GenOptions  # noqa: F401
