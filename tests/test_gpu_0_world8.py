"""The EIGHT-rank cases of tests/test_gpu_partitioned.py -- the p = 3 geometry of BASELINE configs[4] (transposed exchange
with three far bits, k_hypercube_flipsum at P = 8, the (max(3, p) + 1)-slab scratch layout, overlapped exchange, library
driver through the callback communicator) on real HIP slab kernels, eight processes sharing cuda:0, collectives over gloo
staged through the host.

This file sorts FIRST among the GPU tests on purpose.  The GPU serves eight compute processes at a time; a ninth -- the
pytest process itself, once an in-process GPU test has created its context -- puts the run list into time slicing, and
every one of these tests then takes minutes instead of ten seconds (measured: 10 s -> 100-375 s).  Here the parent has not
touched the GPU yet: it only computes CPU oracles and spawns the workers.  (``_run`` skips with that explanation if the
order is ever changed.)"""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_gpu_partitioned as tp  # noqa: E402


@pytest.mark.parametrize("overlap", [False, True])
def test_partitioned_hip_backend_world8(overlap):
    tp.test_partitioned_hip_backend(8, "gloo", overlap)


@pytest.mark.parametrize("overlap", [True, False])
def test_library_driver_equals_python_driver_world8(overlap):
    tp.test_library_driver_equals_python_driver(8, "gloo", overlap)


def test_reference_api_second_order_on_partitioned_hip_operator_world8():
    tp.test_reference_api_second_order_on_partitioned_hip_operator(8, "L12_k200_g1.0")


def test_overlap_premise_is_checked_on_the_device_and_a_violation_repeats_the_run_world8():
    tp.test_overlap_premise_is_checked_on_the_device_and_a_violation_repeats_the_run(8)


def test_library_driver_partial_reorthogonalisation_world8():
    tp.test_library_driver_partial_reorthogonalisation(8, "gloo")


@pytest.mark.parametrize("overlap,env", [(True, {}), (True, tp.ASYNC), (False, {})])
def test_rccl_branch_of_the_library_driver_equals_the_callback_path_world8(overlap, env):
    """the 8-rank geometry of BASELINE configs[4] through the RCCL branch (stand-in RCCL, tests/fake_rccl): 7 sends + 7
    receives per all-to-all, two all-to-alls per mat-vec, on the second communicator"""
    tp.test_rccl_branch_of_the_library_driver_equals_the_callback_path(8, overlap, env)


def test_library_owned_communicators_over_the_rccl_stand_in_collectives_world8():
    tp.test_library_owned_communicators_over_the_rccl_stand_in_collectives(8, {})
