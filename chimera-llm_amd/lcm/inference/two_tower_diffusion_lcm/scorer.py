"""Teacher-forced scoring with the two-tower diffusion LCM on the HIP path.

Same class name, constructor and call signature as the reference's
lcm/inference/two_tower_diffusion_lcm/scorer.py:27-314: every step feeds the gold previous sentence, records the
model's prediction for the next one (never feeding it back) and drops a document after its last sentence;
`hypotheses[i][0].seq` holds the predictions for positions min_context_len .. len_i - 1.  The argument handling and
the step loop are `LCMScorer`'s (lcm/inference/lcm/scorer.py in this package), run on the two-tower generator's model
hooks (prefill through `encode`, one sampling loop per step, reorder as slot-map updates); `noise=` supplies the
N(0, 1) draws as noise[step - min_context_len, document].
"""

from lcm.inference.lcm.scorer import LCMScorer
from lcm.inference.two_tower_diffusion_lcm.generator import TwoTowerDiffusionLCMGenerator


class TwoTowerDiffusionLCMScorer(LCMScorer, TwoTowerDiffusionLCMGenerator):
    """Score by generating in teacher-forcing mode with a two-tower diffusion LCM."""

    def _can_score_in_one_pass(self) -> bool:
        return False           # every sentence has its own sampling loop over the diffusion steps
