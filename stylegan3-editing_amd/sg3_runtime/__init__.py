"""Runtime helpers of the MI355X build that have no counterpart in the reference (which is a single-GPU eager loop):
hipGraph capture of the synthesis forward and the one-process-per-GPU sharding of independent images / frames."""
from .graphed import GraphedReStyleStep, GraphedSynthesis  # noqa: F401
