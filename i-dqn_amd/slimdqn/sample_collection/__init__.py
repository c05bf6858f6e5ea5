"""HBM-resident replay + samplers; mirrors reference ``slimdqn/sample_collection`` (``__init__.py:1-3``)."""
from typing import NewType

ReplayItemID = NewType("ReplayItemID", int)
