import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def witness_d3(oracle):
    """2 transactions at the reference's cfg(test) Merkle depth (src/merkle/constants.rs:22)."""
    return oracle.TxWitness.generate(2, 3, seed=0x5EED)


@pytest.fixture(scope="session")
def witness_d15(oracle):
    """2 transactions at the production Merkle depth 15 (src/merkle/constants.rs:25)."""
    return oracle.TxWitness.generate(2, 15, seed=0x5EED)
