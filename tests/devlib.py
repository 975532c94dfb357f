"""The development build of the library for the tests that compare kernel mappings.

The product library (victor_amd/csrc/libvictor_hip.so) no longer contains the lanes-over-the-batch kernel (a yardstick since
round 3: the cells kernel is ahead of it at every batch size); ``libvictor_hip_dev.so`` (``build_native(dev=True)``,
``make -C victor_amd/csrc dev``) is the product plus that kernel.  ``mapped(fit, "lanes")`` hands out a twin of ``fit`` - the same
host tables - whose device contexts live in the development build; every other mapping runs in the product library."""

import copy
from contextlib import contextmanager

from victor_amd import _native
from victor_amd.build import DEV_OUT, build_native

KERNEL_OF = {"generic": "vk_theory_kernel", "point": "vk_theory_fast_kernel", "cells": "vk_theory_cells_kernel",
             "lanes": "vk_theory_lanes_kernel"}


def dev_library():
    build_native(dev=True)                 # a no-op when the library is current (it travels with the snapshot)
    return _native.load_path(DEV_OUT)


def dev_twin(fit):
    twin = getattr(fit, "_dev_twin", None)
    if twin is None:
        twin = copy.copy(fit)
        twin._engine = None                # its own device contexts ...
        twin._plan = None                  # ... and no cached (engine, options) pair of the original's
        twin._dev_twin = twin
        twin._native_lib = dev_library()
        fit._dev_twin = twin
    return twin


@contextmanager
def mapped(fit, mapping):
    """``with mapped(fit, "lanes") as f:`` - the knob set for the duration, ``f`` the fit to evaluate with."""
    env = "VICTOR_HIP_FORCE_GENERIC" if mapping == "generic" else "VICTOR_HIP_MAPPING"
    target = dev_twin(fit) if mapping == "lanes" else fit
    _native.set_knob(env, "1" if mapping == "generic" else mapping)
    try:
        yield target
    finally:
        _native.set_knob(env, None)
