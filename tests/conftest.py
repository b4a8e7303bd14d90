"""pytest configuration: markers + import paths.

* ``gpu`` marks tests that need a real MI355X (run with ``-m gpu`` on the box).
* The product lives in ``st-gcn-altformer_amd/`` (a drop-in root that provides the
  reference's import names ``model.*`` / ``graph.*`` plus ``stgcn_amd``); the
  oracle lives in ``oracle/`` and is imported by tests only.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "st-gcn-altformer_amd")
for p in (PKG, ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) GPU")
