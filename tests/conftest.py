import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle_exact():
    from oracle import lyra_oracle
    lyra_oracle.build()
    return lyra_oracle.Oracle(mode="exact")


@pytest.fixture(scope="session")
def oracle_double():
    from oracle import lyra_oracle
    lyra_oracle.build()
    return lyra_oracle.Oracle(mode="gemmlowp_double")


@pytest.fixture(scope="session")
def oracle_xnnpack():
    from oracle import lyra_oracle
    lyra_oracle.build()
    return lyra_oracle.Oracle(mode="xnnpack")


@pytest.fixture(scope="session")
def oracle_mixed():
    """Mode "builtin_mixed": TFLite's builtin int8 kernels per operator (the delegate takes only the fp32 operators)."""
    from oracle import lyra_oracle
    lyra_oracle.build()
    return lyra_oracle.Oracle(mode="builtin_mixed")


@pytest.fixture(scope="session")
def oracle_default(oracle_xnnpack):
    """The oracle in the product's default arithmetic mode ("xnnpack": what the reference runs, use_xnn=true)."""
    return oracle_xnnpack
