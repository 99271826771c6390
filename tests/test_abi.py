"""The C-ABI shared library loads and exports every symbol include/sco_hip.h declares."""
import ctypes
import os
import re

import numpy as np
import pytest

from sco_py_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "sco_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sco_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    names = _declared_functions()
    for must in ("sco_qp_create", "sco_qp_load", "sco_qp_set_bounds", "sco_qp_solve", "sco_qp_adaptive_info", "sco_sqp_create",
                 "sco_sqp_load", "sco_sqp_load_target", "sco_sqp_load_vel_limit", "sco_sqp_load_joint_limits", "sco_sqp_set_groups", "sco_sqp_fetch_groups", "sco_sqp_fetch_flags", "sco_sqp_last_rounds", "sco_sqp_solve", "sco_sqp_fetch", "sco_sqp_trace"):
        assert must in names


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.lib_path())
    for name in _declared_functions():
        assert hasattr(lib, name), name


def test_python_binding_table_matches_the_header():
    assert sorted(_lib.ABI.keys()) == _declared_functions()
    _lib.load()      # binds restype/argtypes for all of them


def test_default_settings_are_the_reference_values():
    s = _lib.default_qp_settings()
    # osqp_utils.py:10-15 of the reference + OSQP 0.6 defaults
    assert (s.rho, s.sigma, s.eps_abs, s.eps_rel, s.max_iter) == (0.1, 5e-10, 1e-6, 1e-9, 100000)
    assert (s.alpha, s.check_termination, s.scaling) == (1.6, 25, 10)
    assert (s.adaptive_rho, s.adaptive_rho_interval, s.adaptive_rho_tolerance) == (0, 0, 5.0)   # osqp_utils.py:13
    p = _lib.default_sqp_params()
    # solver.py:17-28
    assert (p.improve_ratio_threshold, p.min_trust_region_size, p.min_approx_improve) == (0.25, 1e-4, 1e-8)
    assert (p.trust_shrink_ratio, p.trust_expand_ratio, p.cnt_tolerance) == (0.1, 1.5, 1e-4)
    assert (p.max_merit_coeff_increases, p.merit_coeff_increase_ratio) == (1, 10.0)
    assert (p.initial_trust_region_size, p.initial_penalty_coeff) == (1.0, 1e3)
    assert (p.compound_penalty, p.duplicate_rows, p.memoize_rounded) == (1, 1, 1)


def test_struct_layouts_match_the_c_side():
    # sizes the C compiler gives the same declarations
    assert ctypes.sizeof(_lib.QpSettings) == 7 * 8 + 6 * 4 + 8
    assert ctypes.sizeof(_lib.SqpParams) == 9 * 8 + 7 * 4 + 4        # 7 ints, padded to the 8-byte alignment
    assert ctypes.sizeof(_lib.TrajoptDesc) == 10 * 4         # r03: + span, n_eq_rows


def test_no_gpu_means_a_loud_failure_not_a_fallback():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible here")
    Pp = np.array([0, 1], dtype=np.int32); Pi = np.array([0], dtype=np.int32)
    with pytest.raises(_lib.ScoHipError) as e:
        _lib.BatchedQP(1, 1, 1, Pp, Pi, Pp, Pi)
    assert e.value.code == -3 and "no CPU fallback" in str(e.value)
    from sco_py_amd.sco_osqp import osqp_utils
    v = osqp_utils.OSQPVar("x")
    with pytest.raises(_lib.ScoHipError):
        osqp_utils.optimize([v], [], [osqp_utils.OSQPQuadraticObj(np.array([v]), np.array([v]), np.array([2.0]))],
                            [osqp_utils.OSQPLinearObj(v, -4.0)], [])
    from sco_py_amd import batch
    with pytest.raises(_lib.ScoHipError):
        batch.TrajOptBatch(2, 3, 4, 1, 1)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "sco_py_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "osqp_ref" not in src.replace("oracle/osqp_ref.c", ""), f


def test_integration_stub_declares_the_same_settings_struct():
    """INTEGRATION.md shows a ctypes stub a maintainer would paste into the reference: its struct must be the
    library's (a shorter one would let sco_qp_default_settings write past its end)."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = text[text.index("class _Settings"):text.index("def _hip_solve")]
    names = re.findall(r'"([a-z_]+)"', stub)
    assert names == [f[0] for f in _lib.QpSettings._fields_]
    header = open(os.path.join(ROOT, "include", "sco_hip.h")).read()
    body = header[header.index("typedef struct sco_qp_settings {"):header.index("} sco_qp_settings;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    assert re.findall(r"\b(?:double|int)\s+([a-z_]+);", body) == names


def test_python_struct_fields_follow_the_header():
    header = open(os.path.join(ROOT, "include", "sco_hip.h")).read()
    for cname, cls in (("sco_sqp_params", _lib.SqpParams), ("sco_trajopt_desc", _lib.TrajoptDesc)):
        body = header[header.index("typedef struct %s {" % cname):header.index("} %s;" % cname)]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in re.findall(r"\b(?:double|int)\s+([a-z_, \n]+);", body):
            names += [v.strip() for v in decl.split(",")]
        assert names == [f[0] for f in cls._fields_], cname
