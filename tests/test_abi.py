"""CPU: the C-ABI shared library loads and exports every symbol include/corrfield.h declares; no compute without a
GPU -- and without a GPU the library must fail loudly, never fall back."""
import ctypes
import re
import subprocess
from pathlib import Path

import pytest

import correrender_amd as ca
from correrender_amd import _lib

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "corrfield.h"


def declared_functions():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(crf_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ca.load_library()
    names = declared_functions()
    assert len(names) >= 15
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/corrfield.h but not exported"
    assert sorted(_lib.SYMBOLS) == names, "python binding table and header diverge"
    assert lib.crf_abi_version() == 5


def test_params_struct_layout_matches_header(tmp_path):
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "corrfield.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu\\n", sizeof(crf_params), offsetof(crf_params, k),'
                   ' offsetof(crf_params, min_ref), offsetof(crf_params, reference_values),'
                   ' offsetof(crf_params, flags)); return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(src), "-o", str(exe)], check=True)
    c_sizes = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    P = _lib.CrfParams
    assert c_sizes == [ctypes.sizeof(P), P.k.offset, P.min_ref.offset, P.reference_values.offset, P.flags.offset]


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "c.c"
    src.write_text('#include "corrfield.h"\nint main(void){crf_params p; p.measure = CRF_PEARSON; return p.measure;}\n')
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-c", f"-I{ROOT / 'include'}", str(src), "-o",
                    str(tmp_path / "c.o")], check=True)


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(ca.CorrFieldError) as e:
        ca.CorrField(0)
    assert e.value.code == 3 and "no CPU fallback" in e.value.message


def test_product_does_not_reference_the_oracle():
    """The product path must not import, link or call anything under oracle/."""
    for p in (ROOT / "correrender_amd").rglob("*"):
        if p.suffix in {".py", ".cpp", ".hip", ".h", ".hpp", ".inc"} or p.name == "Makefile":
            text = p.read_text(errors="ignore")
            assert "liboracle" not in text and "oracle_lib" not in text and "libref_corr" not in text, p
    out = subprocess.run(["ldd", str(ca.library_path())], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_host_only_helpers(oracle):
    """Entry points that need no device: the tiled element count and the KSG colour-range bound psi(cs) - psi(k)."""
    from correrender_amd._lib import load_library
    lib = load_library()
    assert lib.crf_tiled_element_count(19, 10, 6) == 3 * 2 * 2 * 256 and lib.crf_tiled_element_count(0, 1, 1) == 0
    for k, cs in ((3, 64), (1, 2), (30, 1000)):
        assert lib.crf_max_mutual_information_kraskov(k, cs) == oracle.digamma(cs) - oracle.digamma(k)
    assert abs(lib.crf_max_mutual_information_kraskov(3, 64) - 3.228266) < 1e-6      # SURVEY 8(c) probe value
    import math
    assert math.isnan(lib.crf_max_mutual_information_kraskov(0, 64))
