"""pytest configuration: registers the ``gpu`` marker and puts the product package on sys.path.

``-m "not gpu"`` (build container, no GPU): oracle vs golden vectors, host logic, C-ABI symbol table.
``-m gpu`` (MI355X box): parity of the HIP path against the oracle / goldens, through the C-ABI.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "i-dqn_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
