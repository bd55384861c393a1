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


def _header_text():
    import re
    text = open(os.path.join(ROOT, "include", "lcmhip.h")).read()
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def test_header_is_plain_c_and_the_binding_mirrors_its_layouts(tmp_path, hip):
    """The header compiles as strict C99 on its own, and every struct the ctypes binding mirrors has the header's
    size and field offsets; every constant the binding restates has the header's value.  A field added on one side
    only, or reordered, fails here and not as garbage read by the library."""
    import ctypes as C
    import subprocess
    structs = {"lcmhip_tower_desc": hip.TowerDesc, "lcmhip_model_desc": hip.ModelDesc,
               "lcmhip_denoise_params": hip.DenoiseParams, "lcmhip_profile_row": hip.ProfileRow}
    constants = {"LCMHIP_ABI_VERSION": hip.ABI_VERSION, "LCMHIP_STATIC_SCRATCH_BYTES": hip.STATIC_SCRATCH_BYTES,
                 "LCMHIP_F32": hip.F32, "LCMHIP_BF16": hip.BF16,
                 "LCMHIP_NORM_LAYERNORM": hip.NORM_LAYERNORM, "LCMHIP_NORM_RMS": hip.NORM_RMS,
                 "LCMHIP_NORM_FP32": hip.NORM_FP32, "LCMHIP_NORM_UNIT": hip.NORM_UNIT,
                 "LCMHIP_FFN_RELU": hip.FFN_RELU, "LCMHIP_FFN_SWIGLU": hip.FFN_SWIGLU,
                 "LCMHIP_POS_NONE": hip.POS_NONE, "LCMHIP_POS_LEARNED": hip.POS_LEARNED, "LCMHIP_POS_ROPE": hip.POS_ROPE,
                 "LCMHIP_POS_SINE": hip.POS_SINE, "LCMHIP_ORDER_PRE": hip.ORDER["pre"], "LCMHIP_ORDER_POST": hip.ORDER["post"],
                 "LCMHIP_ORDER_NORMFORMER": hip.ORDER["normformer"], "LCMHIP_PRED_SAMPLE": hip.PRED["sample"],
                 "LCMHIP_PRED_EPSILON": hip.PRED["epsilon"], "LCMHIP_PRED_V": hip.PRED["v_prediction"],
                 "LCMHIP_EPI_STORE": hip.EPI_STORE, "LCMHIP_EPI_STORE_F32": hip.EPI_STORE_F32, "LCMHIP_EPI_RELU": hip.EPI_RELU,
                 "LCMHIP_EPI_SWIGLU": hip.EPI_SWIGLU, "LCMHIP_EPI_RESID": hip.EPI_RESID,
                 "LCMHIP_SESSION_KEEP_CONTEXT": hip.SESSION_KEEP_CONTEXT, "LCMHIP_SESSION_SEPARATE_QKV": hip.SESSION_SEPARATE_QKV,
                 "LCMHIP_SESSION_SEPARATE_ROPE": hip.SESSION_SEPARATE_ROPE, "LCMHIP_SESSION_NO_GRAPH": hip.SESSION_NO_GRAPH}
    constants.update({f"LCMHIP_ACT_{name.upper()}": code for name, code in hip.ACT.items()})
    lines = ['#include "lcmhip.h"', "#include <stdio.h>", "int main(void) {"]
    for cname, ct in structs.items():
        lines.append(f'    printf("{cname} %zu\\n", sizeof({cname}));')
        for field, _ in ct._fields_:
            lines.append(f'    printf("{cname}.{field} %zu %zu\\n", offsetof({cname}, {field}), '
                         f'sizeof((({cname}*)0)->{field}));')
    for name in constants:
        lines.append(f'    printf("{name} %ld\\n", (long)({name}));')
    lines += ["    return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    str(src), "-o", str(exe)], check=True, capture_output=True)
    seen = dict(line.split(" ", 1) for line in subprocess.run([str(exe)], check=True, capture_output=True,
                                                               text=True).stdout.splitlines())
    for cname, ct in structs.items():
        assert int(seen[cname]) == C.sizeof(ct), (cname, seen[cname], C.sizeof(ct))
        for field, _ in ct._fields_:
            off, size = map(int, seen[f"{cname}.{field}"].split())
            desc = getattr(ct, field)
            assert (off, size) == (desc.offset, desc.size), (cname, field, off, size, desc.offset, desc.size)
    for name, value in constants.items():
        assert int(seen[name]) == value, (name, seen[name], value)


def test_binding_signatures_follow_the_header(hip):
    """Every prototype of the header is bound with the same number of arguments and the same kind of each (pointer,
    32 / 64-bit integer, size, float), and returns what the header says."""
    import ctypes as C
    import re
    protos = re.findall(r"(const char\s*\*|int)\s+(lcmhip_\w+)\s*\(([^)]*)\)\s*;", _header_text())
    assert sorted(name for _, name, _ in protos) == sorted(hip.EXPORTS)

    def kind_of_c(param):
        param = " ".join(param.split())
        if "*" in param:
            return "pointer"
        words = param.split()[:-1]                           # the type, without the parameter's name
        base = [w for w in words if w != "const"]
        return {"int32_t": "i32", "int": "i32", "uint32_t": "u32", "int64_t": "i64", "size_t": "size", "float": "f32"}[" ".join(base)]

    def kind_of_ctypes(t):
        if t in (C.c_void_p, C.c_char_p) or isinstance(t, type(C.POINTER(C.c_int))):
            return "pointer"
        return {C.c_int32: "i32", C.c_uint32: "u32", C.c_int64: "i64", C.c_size_t: "size", C.c_float: "f32"}[t]

    for ret, name, params in protos:
        fn = getattr(hip.lib, name)
        want = [] if params.strip() in ("", "void") else [kind_of_c(p) for p in params.split(",")]
        got = [kind_of_ctypes(t) for t in fn.argtypes]
        assert got == want, (name, got, want)
        assert fn.restype is (C.c_char_p if "char" in ret else C.c_int), (name, fn.restype)
