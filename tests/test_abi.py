"""The C-ABI library loads and exports every symbol include/lcmhip.h declares; entry points that
need no GPU validate their arguments (no compute calls here)."""

import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lcmhip.h")


@pytest.fixture(scope="module")
def hip():
    from lcm import _hip
    return _hip


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lcmhip_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(hip):
    names = _declared_functions()
    assert len(names) >= 25
    lib = C.CDLL(hip.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # and the Python binding lists the same set
    assert sorted(hip.EXPORTS) == names


def test_header_cites_the_reference(hip):
    text = open(HEADER).read()
    for cite in ("builder.py:215-242", "generator.py:442-466", "ddim.py:358-510", "sonar_normalizer/builder.py:121-137"):
        assert cite in text


def test_version_and_error_channel(hip):
    # the library, the header and the binding agree on the ABI version
    import re
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'lcmhip.h')).read()
    assert hip.lib.lcmhip_version() == hip.ABI_VERSION == int(re.search(r'#define LCMHIP_ABI_VERSION (\d+)', header).group(1))
    desc = hip.ModelDesc(sonar_dim=1024, model_dim=1000, max_seq_len=128, timestep_freq_dim=256, dtype=0, norm_eps=1e-5)
    handle = C.c_void_p()
    rc = hip.lib.lcmhip_model_create(C.byref(desc), C.byref(handle))
    assert rc == -1 and b"model_dim" in hip.lib.lcmhip_last_error()
    with pytest.raises(RuntimeError, match="model_dim"):
        hip.check(rc)


def test_model_table_validation_without_gpu(hip):
    tower = hip.TowerDesc(num_layers=1, num_heads=2, ffn_dim=256, norm_kind=1, ffn_kind=1, pos_kind=2, out_proj_bias=1,
                          rope_theta=10000.0)
    den = hip.TowerDesc(num_layers=1, num_heads=2, ffn_dim=256, norm_kind=1, ffn_kind=1, pos_kind=0, out_proj_bias=1,
                        rope_theta=10000.0)
    desc = hip.ModelDesc(sonar_dim=1024, model_dim=256, max_seq_len=128, timestep_freq_dim=256, dtype=0, norm_eps=1e-5,
                         context=tower, denoiser=den)
    handle = C.c_void_p()
    hip.check(hip.lib.lcmhip_model_create(C.byref(desc), C.byref(handle)))
    try:
        # nothing bound yet: finalize names the first missing tensor
        assert hip.lib.lcmhip_model_finalize(handle) == -4
        assert b"sonar_normalizer.center" in hip.lib.lcmhip_last_error()
        # a session cannot be sized for an unfinished model
        n = C.c_size_t(0)
        assert hip.lib.lcmhip_session_workspace_bytes(handle, 4, 16, 10, 0, C.byref(n)) == -1
        # dtype mismatch and misaligned pointers are rejected at bind time
        shape = (C.c_int64 * 1)(1024)
        assert hip.lib.lcmhip_model_bind_weight(handle, b"sonar_normalizer.center", C.c_void_p(4096), 1, 1, shape) == -1
        assert hip.lib.lcmhip_model_bind_weight(handle, b"sonar_normalizer.center", C.c_void_p(4100), 0, 1, shape) == -1
        assert hip.lib.lcmhip_model_bind_weight(handle, b"sonar_normalizer.center", C.c_void_p(4096), 0, 1, shape) == 0
    finally:
        hip.lib.lcmhip_model_destroy(handle)
    # a denoiser with a positional encoder is a documented gap, reported as such
    den.pos_kind = 2
    desc.denoiser = den
    assert hip.lib.lcmhip_model_create(C.byref(desc), C.byref(handle)) == -3
    # the kernels are written for heads of 128 columns: any other head width is refused by name
    den.pos_kind = 0
    desc.denoiser = den
    tower.num_heads = 4                 # 256 / 4 = 64 columns per head
    desc.context = tower
    assert hip.lib.lcmhip_model_create(C.byref(desc), C.byref(handle)) == -1
    assert b"head_dim must be 128" in hip.lib.lcmhip_last_error()


def test_library_override_and_version_gate(tmp_path, hip):
    """LCMHIP_LIB loads another build of the same ABI; a path that does not exist is an ImportError, never a silent
    fallback to the default library (or to anything on the CPU)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import lcm._hip as h; print(h.LIB_PATH, h.lib.lcmhip_version())"
    copy = tmp_path / "liblcm_other_build.so"
    copy.write_bytes(open(os.path.join(root, "chimera-llm_amd", "lcm", "liblcm_hip.so"), "rb").read())
    env = dict(os.environ, LCMHIP_LIB=str(copy), PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[0] == str(copy) and int(out.stdout.split()[1]) == hip.ABI_VERSION
    env["LCMHIP_LIB"] = str(tmp_path / "missing.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "ImportError" in out.stderr


def test_unknown_session_flags_are_rejected(hip):
    handle = C.c_void_p()          # flags are validated before the model is looked at
    n = C.c_size_t(0)
    assert hip.lib.lcmhip_session_workspace_bytes(handle, 4, 16, 10, 1 << 9, C.byref(n)) == -1
    assert b"flag" in hip.lib.lcmhip_last_error()
