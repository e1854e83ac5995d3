"""The Rust side of the boundary (rust/raytracer-hip-sys, rust/src/gpu.rs) ships uncompiled -- there is no Rust toolchain
in the image -- so tools/check_rust_abi.py checks it against include/rt_abi.h and the ctypes / numpy mirror the GPU tests
call through.  These tests run the checker, and make sure it does notice the drift it is there to catch."""
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _checker():
    spec = importlib.util.spec_from_file_location("check_rust_abi", os.path.join(ROOT, "tools", "check_rust_abi.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_rust_declarations_match_the_header(capsys):
    assert _checker().main() == 0, capsys.readouterr().out


@pytest.mark.parametrize("what, edit", [
    ("field order", lambda s: s.replace("pub vclass: u32,\n    pub prim: u32,", "pub prim: u32,\n    pub vclass: u32,")),
    ("field type", lambda s: s.replace("pub parent: i32,\n    pub vclass", "pub parent: u32,\n    pub vclass")),
    ("missing field", lambda s: s.replace("    pub rays_reference: u64,\n", "")),
    ("constant value", lambda s: s.replace("RT_E_LIMIT: c_int = -6", "RT_E_LIMIT: c_int = -7")),
    ("missing constant", lambda s: s.replace("pub const RT_DITHER_BAYER16: u32 = 4;\n", "")),
    ("missing function", lambda s: re.sub(r"    pub fn rt_check\([^;]*;\n", "", s)),
    ("argument type", lambda s: s.replace("dither: u32,\n        out_u8: *mut u8", "dither: u32,\n        out_u8: *mut f32")),
    ("argument missing", lambda s: s.replace("        remaining: u32,\n", "", 1)),
    ("repr(C) dropped", lambda s: s.replace("#[repr(C)]\n#[derive(Clone, Copy, Debug, Default, PartialEq)]\npub struct RtNode",
                                            "#[derive(Clone, Copy, Debug, Default, PartialEq)]\npub struct RtNode")),
])
def test_checker_notices_drift(tmp_path, capsys, what, edit):
    chk = _checker()
    src = open(chk.RUST).read()
    changed = edit(src)
    assert changed != src, f"the edit for '{what}' did not apply"
    p = tmp_path / "lib.rs"
    p.write_text(changed)
    chk.RUST = str(p)
    assert chk.main() == 1, what
    assert "MISMATCH" in capsys.readouterr().out


def test_checker_notices_unknown_names_in_gpu_rs(tmp_path, capsys):
    chk = _checker()
    src = open(chk.GPU_RS).read()
    p = tmp_path / "gpu.rs"
    p.write_text(src.replace("n.first_tri = first_tri;", "n.first_tri = first_tri; let _ = RT_NODE_NO_SUCH_FLAG;")
                 .replace("RtCsgRoot { first, count:", "RtCsgRoot { first, n_nodes:"))
    chk.GPU_RS = str(p)
    assert chk.main() == 1
    out = capsys.readouterr().out
    assert "RT_NODE_NO_SUCH_FLAG" in out and "n_nodes" in out


def test_gpu_rs_mirrors_every_flatten_step_of_the_cpp_host():
    """Function for function: each step of the C++ flatten that the GPU tests exercise has its twin in gpu.rs, named in
    the twin's doc comment."""
    rs = open(os.path.join(ROOT, "rust", "src", "gpu.rs")).read()
    cpp = open(os.path.join(ROOT, "raytracer_amd", "csrc", "host", "rt_host.cpp")).read()
    for fn in ["xform_index", "tables", "material_slow", "fill_tri", "emit_shape", "emit_csg", "emit_group", "run",
               "make_rt_camera", "true_psb", "KeyBuilder", "ClassTable", "geometry_run"]:
        assert re.search(rf"\b{fn}\b", cpp), fn
        assert re.search(rf"rt_host\.cpp: `{fn}", rs) or re.search(rf"rt_host\.cpp: `?{fn}", rs), fn
