"""The C-ABI shared library loads on a CPU-only box and exports every symbol
declared in include/hyperion_amd.h; the ctypes mirror has the C layout; the
engine refuses to run without a GPU (no silent CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

import hyperion_amd
from hyperion_amd import _abi
from hyperion_amd.benchmark import make_benchmark_problem
from hyperion_amd.build import build_extension

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hyperion_amd.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hyp_\w+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    lib = C.CDLL(build_extension())
    names = declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "missing export %s" % n
    assert sorted(hyperion_amd.engine.EXPORTS) == names
    assert hyperion_amd.load_library().hyp_abi_version() == 2


def test_ctypes_layout_matches_the_header(tmp_path):
    structs = {"hyp_dust_desc": _abi.DustDesc, "hyp_source_desc": _abi.SourceDesc, "hyp_grid_desc": _abi.GridDesc,
               "hyp_config": _abi.Config, "hyp_peeled_desc": _abi.PeeledDesc, "hyp_problem": _abi.ProblemDesc,
               "hyp_iter_stats": _abi.IterStats}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "hyperion_amd.h"', '#include "hyp_oracle.h"', 'int main(void){']
    for cname, ct in structs.items():
        oname = cname.replace("hyp_", "orc_")
        lines.append('printf("%s %%zu %%zu\\n", sizeof(%s), sizeof(%s));' % (cname, cname, oname))
        for f, _ in ct._fields_:
            lines.append('printf("%s.%s %%zu %%zu\\n", offsetof(%s,%s), offsetof(%s,%s));' % (cname, f, cname, f, oname, f))
    lines.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split("\n")
    got = {l.split()[0]: (int(l.split()[1]), int(l.split()[2])) for l in out if l.strip()}
    for cname, ct in structs.items():
        assert got[cname] == (C.sizeof(ct), C.sizeof(ct)), cname
        for f, _ in ct._fields_:
            off = getattr(ct, f).offset
            assert got["%s.%s" % (cname, f)] == (off, off), "%s.%s" % (cname, f)


def test_engine_fails_loudly_without_a_gpu():
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("GPU present")
    with pytest.raises(hyperion_amd.EngineError, match="no HIP device"):
        hyperion_amd.Engine(make_benchmark_problem(4))


def test_missing_library_is_an_error(tmp_path):
    with pytest.raises(hyperion_amd.EngineError, match="missing"):
        hyperion_amd.load_library(str(tmp_path / "nope.so"))


def test_product_never_touches_the_oracle():
    """The product path must not import, link or read anything under oracle/."""
    pkg = os.path.join(ROOT, "hyperion_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in txt and "hyp_oracle" not in txt and "libhyp_oracle" not in txt, f


def test_every_option_name_used_by_tools_tests_and_bench_is_known_to_the_engine():
    """a pruned option left tools reading names the engine no longer answers.  Source-level check (runs without
    a GPU): every literal name passed to get_option / set_option under tools/, tests/ and bench.py appears in
    hyp_get_option / hyp_set_option of hyp_engine.hip."""
    import glob
    import re
    src = open(os.path.join(ROOT, "hyperion_amd", "csrc", "hyp_engine.hip")).read()
    body = {}
    for which in ("set", "get"):
        start = src.index("int hyp_%s_option(" % which)
        body[which] = src[start:src.index("\n}\n", start)]
    known = {w: set(re.findall(r'n == "([a-z_0-9]+)"', body[w])) for w in body}
    known["get"].update("last_walk_why%d" % i for i in range(8))
    files = glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "tests", "*.py")) + [os.path.join(ROOT, "bench.py")]
    bad = []
    for f in files:
        text = open(f).read()
        for which, name in re.findall(r'\b(get|set)_option\(\s*"([a-z_0-9]+)"', text):
            if name not in known[which]:
                bad.append((os.path.relpath(f, ROOT), which, name))
        for blob in re.findall(r'dict\(((?:\s*[a-z_0-9]+=[^,()]+,?)+)\)', text) if "set_option(k, v)" in text else []:
            for name in re.findall(r'([a-z_0-9]+)=', blob):
                if name.startswith(("tile_", "vt_", "ot_", "at_", "lucy_mode")) and name not in known["set"]:
                    bad.append((os.path.relpath(f, ROOT), "set", name))
    assert not bad, bad
    assert len(known["get"]) > 40 and len(known["set"]) > 20
