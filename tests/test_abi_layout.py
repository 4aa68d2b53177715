"""The ctypes mirrors of the public structs (vision-transformer-opencl_amd/binding.py) against what a C compiler makes of
include/*.h: size of every struct and offset of every field.  A field added to a header and not to its mirror (or the other way
round) shifts everything behind it silently -- pointers land in ints -- so the layout is compared, not assumed.  No GPU."""
import ctypes as C
import os
import subprocess

from vit_amd import binding as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MIRRORS = {   # C type -> (header, ctypes mirror)
    "vit_config": ("vit_types.h", B.CConfig),
    "Network": ("vit_types.h", B.CNetwork),
    "ImageData": ("vit_types.h", B.CImageData),
    "vit_engine_options": ("vit_engine.h", B.COptions),
    "vit_stage_times": ("vit_engine.h", B.CStageTimes),
    "vithip_device_info": ("vit_hip_kernels.h", B.CDeviceInfo),
    "vithip_gemm_args": ("vit_hip_kernels.h", B.CGemmArgs),
    "vithip_gemm_bf16_args": ("vit_hip_kernels.h", B.CGemmBf16Args),
    "vit_weight_image": ("vit_io.h", B.CWeightImage),
}


def test_ctypes_mirrors_have_the_layout_of_the_headers(tmp_path):
    lines = ["#include <stdio.h>", "#include <stddef.h>"]
    for header in sorted({h for h, _ in MIRRORS.values()}):
        lines.append(f'#include "{header}"')
    lines.append("int main(void) {")
    for ctype, (_, mirror) in MIRRORS.items():
        lines.append(f'    printf("{ctype} %zu\\n", sizeof({ctype}));')
        for name, *_ in mirror._fields_:
            lines.append(f'    printf("{ctype}.{name} %zu\\n", offsetof({ctype}, {name}));')
    lines += ["    return 0;", "}"]
    src, exe = tmp_path / "layout.c", tmp_path / "layout"
    src.write_text("\n".join(lines))
    # a field the mirror names and the header does not have is a compile error here -- which is the point
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True, capture_output=True, text=True)
    out = dict(ln.split() for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for ctype, (_, mirror) in MIRRORS.items():
        assert int(out[ctype]) == C.sizeof(mirror), (ctype, out[ctype], C.sizeof(mirror))   # a header field missing from the mirror shows here
        for name, *_ in mirror._fields_:
            assert int(out[f"{ctype}.{name}"]) == getattr(mirror, name).offset, (ctype, name)
