"""The product package never imports the oracle (task rule: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may) and fails loudly without its HIP library."""
import ast
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "learning-continuous-implicit-representation-for-near-periodic-patterns_amd")


def _imports(path):
    tree = ast.parse(open(path).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            for a in node.names:
                yield a.name, node.lineno
        elif isinstance(node, ast.ImportFrom) and node.module:
            yield node.module, node.lineno


def test_package_is_oracle_free():
    for fn in sorted(os.listdir(PKG)):
        if fn.endswith(".py"):
            for mod, line in _imports(os.path.join(PKG, fn)):
                assert mod.split(".")[0] != "oracle", f"{fn}:{line} imports {mod}"


def test_bench_touches_oracle_only_in_cpu_baseline():
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    for node in tree.body:
        names = [n for n in ast.walk(node) if isinstance(n, (ast.Import, ast.ImportFrom))]
        for n in names:
            mods = [a.name for a in n.names] if isinstance(n, ast.Import) else [n.module or ""]
            if any(m.split(".")[0] == "oracle" for m in mods):
                assert isinstance(node, ast.FunctionDef) and node.name == "cpu_baseline", f"oracle imported outside cpu_baseline (line {n.lineno})"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import importlib
    import npp_amd._lib as L
    monkeypatch.setattr(L, "LIB_PATHS", {256: str(tmp_path / "nope.so"), 512: str(tmp_path / "nope_w512.so")})
    monkeypatch.setattr(L, "_LIBS", {})
    try:
        for w in (256, 512):
            try:
                L.lib(w)
                raise AssertionError("lib() must raise when the library is missing")
            except L.NppError as e:
                assert "no CPU fallback" in str(e)
        try:
            L.lib(384)
            raise AssertionError("lib() must raise for a width without a fused build")
        except L.NppError as e:
            assert "width 384" in str(e)
    finally:
        importlib.reload(L)
