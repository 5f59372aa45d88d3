"""Stack of message passing steps (mirror of /root/reference/src/mpn/message_passing_stack.py:9-71)."""
from typing import Any, Dict, Optional

import torch.nn as nn

from amber_amd.mpn.message_passing_block import MessagePassingBlock


class MessagePassingStack(nn.Module):
    def __init__(self, latent_dimension: int, stack_config: Dict[str, Any]):
        super().__init__()
        self._num_steps: int = stack_config.get("num_steps")
        self._num_step_repeats: int = stack_config.get("num_step_repeats", 1)
        self._residual_connections: Optional[str] = stack_config.get("residual_connections")
        self._latent_dimension: int = latent_dimension
        self._message_passing_steps = nn.ModuleList(
            [MessagePassingBlock(stack_config=stack_config, latent_dimension=latent_dimension) for _ in range(self._num_steps)]
        )

    @property
    def num_steps(self) -> int:
        return self._num_steps

    @property
    def num_step_repeats(self) -> int:
        return self._num_step_repeats

    @property
    def latent_dimension(self) -> int:
        return self._latent_dimension

    def param_list(self):
        out = []
        for block in self._message_passing_steps:
            out += block.param_list()
        return out

    def forward(self, graph) -> None:
        """In place on a latent graph; separate weights per step, shared across repeats (stack.py:58-60).
        MessagePassingBase.forward does not come through here: it runs the whole stack in one fused call."""
        for block in self._message_passing_steps:
            for _ in range(self._num_step_repeats):
                block(graph=graph)

    def __repr__(self):
        first = f" first_step={self._message_passing_steps[0]}\n" if len(self._message_passing_steps) else ""
        return (f"{self.__class__.__name__}(\n num_message_passing_steps={self.num_steps},\n"
                f" num_step_repeats={self._num_step_repeats},\n{first}")
