"""State bag handed between the generator and the model during incremental decoding.

The reference's LCMIncrementalStateBag (lcm/nn/incremental_state.py:14-43) keeps one
FullAttentionState per attention module.  Here the key/value tensors live in the HIP session's
workspace, so the bag only carries the step counter and the handle of that session.
"""

from typing import Optional


class LCMIncrementalStateBag:
    def __init__(self, max_num_steps: int, *, capacity_increment: Optional[int] = 16) -> None:
        self.max_num_steps = int(max_num_steps)
        self.capacity_increment = capacity_increment
        self.step_nr = 0
        self.session = None          # lcm.models.two_tower_diffusion_lcm.builder.DeviceSession

    def increment_step_nr(self, value: int = 1) -> None:
        if self.step_nr + value > self.max_num_steps:
            raise ValueError(
                f"The current step number ({self.step_nr}) must be less than or equal to the maximum "
                f"({self.max_num_steps}) after incrementing by {value}.")
        self.step_nr += value

    def reorder(self, new_order) -> None:
        """Keep only rows `new_order` (ascending indices into the current batch)."""
        if self.session is not None:
            self.session.reorder(new_order)
