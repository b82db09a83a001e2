"""CPU tests (-m "not gpu"): the C-ABI library loads and exports every symbol
include/albatross_amd.h declares; the host mirror flattens covariance
functions correctly.  No compute calls (there is no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import albatross_amd as ab
from albatross_amd import _capi as capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "albatross_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(agp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/albatross_amd.h but not exported"
    assert sorted(n for n, _, _ in capi.EXPORTS) == names


def _exported(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
    return [ln.split()[-1] for ln in out.splitlines() if ln.strip()]


def test_exported_symbols_are_exactly_the_declared_ones():
    """-fvisibility=hidden + AGP_API: the product library exports the entry points of include/albatross_amd.h and
    nothing else of its own - no internal C++ symbol, no agp_debug_* probe (those live in libalbatross_amd_debug.so)."""
    syms = _exported(capi.lib_path())
    ours = sorted(s for s in syms if s.startswith("agp_"))
    assert ours == declared_symbols()
    assert not [s for s in syms if "agp" in s and not s.startswith("agp_")], "internal C++ symbols are exported"
    assert not [s for s in ours if s.startswith("agp_debug")]
    dbg = _exported(os.path.join(os.path.dirname(capi.lib_path()), "libalbatross_amd_debug.so"))
    assert len([s for s in dbg if s.startswith("agp_debug_")]) >= 10
    assert set(ours) <= set(dbg)


def test_status_strings_and_device_count_without_gpu():
    lib = capi.load()
    assert lib.agp_status_string(capi.AGP_OK) == b"ok"
    assert b"positive definite" in lib.agp_status_string(capi.AGP_ERR_NOT_POSITIVE_DEFINITE)
    assert lib.agp_device_count() >= 0


def test_kernel_create_validates_programs():
    lib = capi.load()
    cov = ab.SquaredExponential(1., 1.) + ab.IndependentNoise(0.1)
    nodes = cov.program_nodes()
    arr = (capi.KernelNode * len(nodes))(*nodes)
    h = C.c_void_p()
    assert lib.agp_kernel_create(arr, len(nodes), C.byref(h)) == capi.AGP_OK
    lib.agp_kernel_destroy(h)
    # a SUM with one operand is malformed
    bad = (capi.KernelNode * 2)(nodes[0], nodes[2])
    assert lib.agp_kernel_create(bad, 2, C.byref(h)) == capi.AGP_ERR_INVALID_ARGUMENT
    # two leaves and no operator leave two values on the stack
    bad = (capi.KernelNode * 2)(nodes[0], nodes[1])
    assert lib.agp_kernel_create(bad, 2, C.byref(h)) == capi.AGP_ERR_INVALID_ARGUMENT


def test_postfix_program_of_composed_covariance():
    cov = ab.ScalingTerm(type("F", (ab.ScalingFunction,), {"_call_impl": lambda self, c: np.ones(len(c))})()) \
        * ab.Constant(5.) + ab.measurement_only(ab.IndependentNoise(1.75)) \
        + ab.Exponential(1.1, 1., ab.AngularDistance()) * ab.SquaredExponential(5835., 13.9, ab.RadialDistance())
    ops = [n.op for n in cov.program_nodes()]
    assert ops == [capi.OP_SCALING, capi.OP_CONSTANT, capi.OP_PRODUCT, capi.OP_INDEPENDENT_NOISE,
                   capi.OP_MEASUREMENT_ONLY, capi.OP_SUM, capi.OP_EXPONENTIAL, capi.OP_SQUARED_EXPONENTIAL,
                   capi.OP_PRODUCT, capi.OP_SUM]
    assert cov.get_name().startswith("(((F*constant)+measurement[independent_noise])+")


def test_parameter_handling_mirrors_reference_names():
    cov = ab.SquaredExponential(3.5, 5.7) + ab.measurement_only(ab.IndependentNoise(1.0))
    assert cov.get_params() == {"squared_exponential_length_scale": 3.5, "sigma_squared_exponential": 5.7,
                                "sigma_independent_noise": 1.0}
    cov.set_param_values({"sigma_independent_noise": 0.25})
    assert cov.get_params()["sigma_independent_noise"] == 0.25
    assert cov.program_nodes()[1].params[0] == 0.25
    with pytest.raises(KeyError):
        cov.set_param("no_such_param", 1.)
    with pytest.raises(TypeError):
        ab.SquaredExponential(1., 1., ab.AngularDistance())  # static_assert radial.hpp:138-141


def test_no_fallback_when_library_missing(monkeypatch, tmp_path):
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "lib_path", lambda: str(tmp_path / "libalbatross_amd.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        capi.load()


def test_product_path_never_imports_oracle():
    pkg = os.path.join(ROOT, "albatross_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower(), f"{f} mentions the oracle"


def test_host_cxx_under_address_and_ub_sanitizers():
    """examples/Makefile `asan`: the sharded-fit schedule (albatross_amd/csrc/shard_sched.hip, plain C++, compiled into
    the test binary with -fsanitize=address,undefined) driven with naive block operations on one rank and through the
    forced multi-rank path, plus the header-only host layer; no GPU involved."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    ex = os.path.join(ROOT, "examples")
    subprocess.check_call(["make", "-s", "-C", ex, "asan"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    run = subprocess.run([os.path.join(ex, "host_sanitize_check_asan")], env=env, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "host_sanitize_check ok" in run.stdout and "ERROR" not in run.stderr
