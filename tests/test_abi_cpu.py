"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/*.h declares,
and refuses to run (loudly) without a GPU — there is no CPU fallback in the product."""
import os
import re

import numpy as np
import pytest

from helpers import F, REPO, hip_lib, oracle_lib


def declared_symbols():
    text = open(os.path.join(REPO, "include", "avian_mi355x.h")).read()
    return sorted(set(re.findall(r"AVN_FN\((\w+)\)\s*\(", text)))


def test_header_symbols_all_exported_by_both_libraries():
    syms = declared_symbols()
    assert len(syms) >= 26 and set(syms) == set(F.ABI_SYMBOLS)
    import ctypes
    for lib in (hip_lib(), oracle_lib()):
        dll = ctypes.CDLL(lib.path)
        for s in syms:
            assert hasattr(dll, lib.prefix + s), f"{lib.path} does not export {lib.prefix}{s}"


def test_product_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the -m gpu tests")
    with pytest.raises(F.AvnError) as e:
        F.World(hip_lib(), F.default_config(32))
    assert e.value.status == 5 and "no CPU fallback" in str(e.value)


def test_product_does_not_link_or_import_the_oracle():
    import subprocess
    out = subprocess.run(["ldd", hip_lib().path], capture_output=True, text=True).stdout
    assert "oracle" not in out
    for root, _, files in os.walk(os.path.join(REPO, "avian_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                src = open(os.path.join(root, f)).read()
                assert "liboracle" not in src and "avo_" not in src and "oracle/" not in src, f"{f} references the oracle"


def test_pair_key_and_config_struct_layout():
    lib = hip_lib()
    assert lib.pair_key(5, 3) == (3 << 32) | 5 == oracle_lib().pair_key(3, 5)
    import ctypes
    assert ctypes.sizeof(F.avn_config) == 128
    assert ctypes.sizeof(F.avn_pair) == 24 and F.PAIR_DTYPE.itemsize == 24


def test_host_constraint_graph_matches_oracle_graph_cpu():
    """Host-side colouring (integer work, runs without a GPU): product C++ vs oracle restatement, incl. pop."""
    rng = np.random.default_rng(0)
    n, m = 300, 3000
    go, gh = F.ConstraintGraph(oracle_lib()), F.ConstraintGraph(hip_lib())
    static = rng.random(n) < 0.1
    live = []
    for i in range(m):
        a = int(rng.integers(0, n)); b = int((a + 1 + rng.integers(0, n - 1)) % n)
        if static[a] and static[b]:
            continue
        co = go.push(i, a, b, bool(static[a]), bool(static[b]))
        ch = gh.push(i, a, b, bool(static[a]), bool(static[b]))
        assert co == ch
        live.append(i)
        if rng.random() < 0.3 and live:
            h = live.pop(int(rng.integers(0, len(live))))
            go.pop(h); gh.pop(h)
    oo, ho = go.lists(); oh, hh = gh.lists()
    assert np.array_equal(oo, oh) and np.array_equal(ho, hh)
    assert oo[24] - oo[23] >= 0


def test_every_ctypes_struct_has_the_size_the_c_compiler_gives_the_header_struct(tmp_path):
    """The binding mirrors the header by hand: compile the header with gcc (as C) and compare sizeof() of every avn_* struct
    the binding declares.  Catches a field added on one side only (e.g. avn_timers.island_blocks)."""
    import ctypes
    import inspect
    import subprocess
    names = sorted(n for n, c in inspect.getmembers(F, inspect.isclass) if issubclass(c, ctypes.Structure) and n.startswith("avn_"))
    assert len(names) >= 15
    header = os.path.join(REPO, "include", "avian_mi355x.h")
    declared = open(header).read()
    names = [n for n in names if f"}} {n};" in declared or f"struct {n} " in declared]
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "avian_mi355x.h"\nint main(void) {\n' +
                   "".join(f'  printf("{n} %zu\\n", sizeof({n}));\n' for n in names) + "  return 0;\n}\n")
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    got = dict(line.split() for line in out.strip().splitlines())
    assert len(got) >= 15
    for n in names:
        assert int(got[n]) == ctypes.sizeof(getattr(F, n)), f"{n}: header {got[n]} B, binding {ctypes.sizeof(getattr(F, n))} B"


def test_generated_rust_bindings_are_current_and_complete():
    """integration/rust/avian_mi355x-sys/src/lib.rs (what a maintainer's Rust shim links against) is generated from the header:
    it must be up to date, declare every exported symbol, and give every struct exactly the fields of the ctypes mirror, in order."""
    import ctypes
    import inspect
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "gen_rust_bindings.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    rs = open(os.path.join(REPO, "integration", "rust", "avian_mi355x-sys", "src", "lib.rs")).read()
    for sym in declared_symbols():
        assert re.search(rf"pub fn avn_{sym}\(", rs), f"avn_{sym} missing from the Rust declarations"
    assert len(re.findall(r"pub fn avn_", rs)) == len(declared_symbols())
    structs = {n: c for n, c in inspect.getmembers(F, inspect.isclass) if issubclass(c, ctypes.Structure) and n.startswith("avn_")}
    checked = 0
    for n, c in structs.items():
        m = re.search(rf"pub struct {n} \{{(.*?)\n\}}", rs, flags=re.S)
        if not m:
            continue
        rust_fields = re.findall(r"pub (\w+):", m.group(1))
        assert rust_fields == [f[0] for f in c._fields_], f"{n}: Rust fields {rust_fields} != binding fields {[f[0] for f in c._fields_]}"
        checked += 1
    assert checked >= 15
