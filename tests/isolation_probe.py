"""NOT collected by the suite (the file name does not match test_*.py): bodies that tests/test_host_cpu.py hands to
conftest.run_isolated() to prove that a child which dies natively is reported as an ordinary failure of the calling test."""
import os

import pytest


def test_probe_aborts():
    os.abort()


def test_probe_passes():
    assert 1 + 1 == 2


def test_probe_skips():
    pytest.skip("probe")
