"""The export / header checks of test_abi.py again under the `gpu` marker, so that they also run against the library
as it arrives on the GPU box (the CPU run covers the container's build)."""

import pytest

import test_abi

pytestmark = pytest.mark.gpu

hip = test_abi.hip


def test_every_declared_symbol_is_exported_on_the_gpu_box(hip):
    test_abi.test_every_declared_symbol_is_exported(hip)


def test_error_channel_and_model_table_validation_on_the_gpu_box(hip):
    test_abi.test_version_and_error_channel(hip)
    test_abi.test_model_table_validation_without_gpu(hip)
