"""rectipy_amd -- MI355X-native engine for RectiPy's time-stepped integration + BPTT path.

Same front-end names as the reference (``rectipy/__init__.py:34``): ``Network``, ``Observer``;
the arithmetic runs in hand-written gfx950 kernels behind the C ABI of ``include/rectipy_hip.h``.
"""
__version__ = "0.1.0"

from .network import Network, PerformanceWarning
from .observer import Observer
from .nodes import InstantNode, MultiSpikeResetNet, RateNet, Spike, SpikeNet, SpikeResetNet
from .edges import Linear
from .feedback import FeedbackNetwork
from .connectivity import (circular_connectivity, input_connections, line_connectivity, normalize,
                           random_connectivity, wta_score)

__all__ = ["Network", "PerformanceWarning", "FeedbackNetwork", "Observer", "RateNet", "SpikeNet", "SpikeResetNet", "MultiSpikeResetNet", "Spike", "InstantNode", "Linear",
           "random_connectivity", "circular_connectivity", "line_connectivity", "input_connections", "normalize",
           "wta_score"]
