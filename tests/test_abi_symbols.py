"""CPU checks of the drop-in boundary: both shared libraries load here (no GPU), export every
function include/*.h declares, agree with the headers on layouts and flags, and refuse to compute
without a device instead of falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"_[a-z0-9_]+)\s*\(", text)))


def test_librt_hip_exports_every_declared_symbol():
    from raytracer_amd import _abi

    lib = _abi.load()
    names = _declared("rt_hip.h", "rt")
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"librt_hip.so does not export {n}"
    assert sorted(_abi.SIGNATURES) == names, "ctypes table and rt_hip.h disagree"


def test_librt_host_exports_every_declared_symbol():
    from raytracer_amd import host

    lib = host.load()
    names = _declared("rt_host.h", "rth")
    for n in names:
        assert hasattr(lib, n), f"librt_host.so does not export {n}"
    assert sorted(host.SIGNATURES) == names


def test_flag_values_match_header_and_oracle():
    import oracle_lib as O
    from raytracer_amd import _abi

    text = open(os.path.join(ROOT, "include", "rt_types.h")).read()
    defs = {m.group(1): int(m.group(2), 16) for m in re.finditer(r"#define (RT_[A-Z0-9_]+) (0x[0-9a-fA-F]+)u", text)}
    for name, val in defs.items():
        assert getattr(_abi, name) == val
    assert (O.ACCUM_F32, O.DISPATCH_FULL, O.SHADE_CHECKER, O.COMBINE_SUM) == (
        defs["RT_ACCUM_F32"], defs["RT_DISPATCH_FULL"], defs["RT_SHADE_CHECKER"], defs["RT_COMBINE_SUM"])


def test_pod_layouts():
    import raytracer_amd as R

    assert R.SPHERE_DTYPE.itemsize == 32 and R.MATERIAL_DTYPE.itemsize == 32 and R.CAMERA_DTYPE.itemsize == 160
    f = R.CAMERA_DTYPE.fields
    assert (f["pixel_to_world"][1], f["pos"][1], f["samples_per_pixel"][1], f["frames_to_render"][1], f["current_frame"][1]) == (64, 128, 148, 152, 156)
    assert C.sizeof(R._abi.RtCounters) == 112


def test_no_device_is_an_error_not_a_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import raytracer_amd as R

    with pytest.raises(R.RtError) as e:
        R.Raytracer(64, 64)
    assert e.value.code == R._abi.RT_ERR_NO_DEVICE
    assert "no CPU path" in str(e.value)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under raytracer_amd/ may reference it."""
    pkg = os.path.join(ROOT, "raytracer_amd")
    for base, _dirs, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h", ".hpp")) or fn == "Makefile":
                text = open(os.path.join(base, fn), errors="replace").read()
                assert "oracle_lib" not in text and "rt_oracle" not in text and "librt_oracle" not in text, fn


def test_headers_are_plain_c_and_a_c_caller_links(tmp_path):
    """include/*.h is the drop-in boundary: it must parse as strict C99 (what cgo / bindgen / a C host sees) and a
    C program that calls it must link against the shared libraries (no compute call is made without a GPU)."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc = os.path.join(root, "include")
    src = tmp_path / "caller.c"
    src.write_text(
        '#include "rt_hip.h"\n#include "rt_host.h"\n#include <stdio.h>\n'
        "int main(void) {\n"
        "  rt_ctx* ctx = 0;\n"
        "  /* width 0 is rejected before any device is touched */\n"
        "  int rc = rt_create(&ctx, 0, 0u, 0u, RT_ACCUM_F32 | RT_MATH_STRICT);\n"
        "  uint32_t bits[3];\n"
        "  int rc2 = rth_time_table(0u, 3u, bits);\n"
        '  printf("%d %d %08x %s\\n", rc, rc2, (unsigned)bits[2], rt_version());\n'
        "  return (rc == RT_ERR_INVALID && rc2 == RT_OK && ctx == 0) ? 0 : 1;\n}\n")
    exe = tmp_path / "caller"
    lib = os.path.join(root, "raytracer_amd")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", f"-I{inc}", str(src), "-o", str(exe),
                    f"-L{lib}", "-lrt_host", "-lrt_hip", f"-Wl,-rpath,{lib}"], check=True, capture_output=True, timeout=120)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "3cf5c28f" in out.stdout  # 3 x 0.01f accumulated (time.rs:72-77)
