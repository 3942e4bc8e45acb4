"""Stand-in for torchvision (absent in this image): routes the three ops the reference
imports to the oracle's restatement in oracle/ops_ref.py. TEST INFRASTRUCTURE ONLY."""
__version__ = "0.16.2"
from . import ops  # noqa
