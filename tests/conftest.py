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


def pytest_sessionstart(session):
    """A checkout that arrives without the built extension (it is git-ignored) compiles it once, up front: the ABI
    test and every GPU test load it, and there is no CPU path to fall back to.  A failed build fails those tests."""
    lib = os.path.join(ROOT, "i-dqn_amd", "libidqn_hip.so")
    if not os.path.exists(lib) and os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        try:
            import __graft_entry__

            __graft_entry__.build()
        except Exception as e:  # the tests that need the library will say what is missing
            print(f"[conftest] building the HIP extension failed: {e}", file=sys.stderr)
