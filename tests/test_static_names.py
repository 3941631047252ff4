"""CPU: every name a function of the product, the bench, the driver hooks or a test reads resolves to a local, an enclosing
scope, a module-level binding or a builtin.  The GPU tests cannot run in the build container, so a name that is used but never
imported there (round 3's `ocr_host` in tests/test_gpu_round3.py) would otherwise surface only on the driver's GPU box."""
import builtins
import re
import symtable
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
FILES = sorted([*ROOT.glob("tests/*.py"), *ROOT.glob("tests/golden/*.py"), *ROOT.glob("rapiddoc_amd/*.py"), *ROOT.glob("oracle/*.py"),
                *ROOT.glob("tools/*.py"), ROOT / "bench.py", ROOT / "__graft_entry__.py"])
MODULE_DUNDERS = {"__file__", "__name__", "__doc__", "__package__", "__spec__", "__loader__", "__builtins__", "__path__", "__class__"}


def _tables(t):
    yield t
    for c in t.get_children():
        yield from _tables(c)


def undefined_names(path: Path):
    top = symtable.symtable(path.read_text(), str(path), "exec")
    module_names = {s.get_name() for s in top.get_symbols() if s.is_assigned() or s.is_imported() or s.is_namespace()}
    for t in _tables(top):                                   # `global x; x = ...` inside a function binds at module level
        module_names |= {s.get_name() for s in t.get_symbols() if s.is_declared_global() and s.is_assigned()}
    known = module_names | set(dir(builtins)) | MODULE_DUNDERS
    bad = []
    for t in _tables(top):
        for s in t.get_symbols():
            if not s.is_referenced():
                continue
            reads_module_scope = s.is_global() or (t is top and not (s.is_assigned() or s.is_imported() or s.is_namespace()))
            if reads_module_scope and s.get_name() not in known:
                bad.append(f"{path.relative_to(ROOT)}:{t.get_lineno()} {t.get_name()}(): name '{s.get_name()}' is not defined")
    return bad


@pytest.mark.parametrize("path", FILES, ids=lambda p: str(p.relative_to(ROOT)))
def test_no_undefined_names(path):
    src = path.read_text()
    assert not re.search(r"^\s*from\s+\S+\s+import\s+\*", src, re.M), "star imports defeat the static name check"
    assert undefined_names(path) == []


def test_the_checker_sees_the_round3_failure(tmp_path):
    """The exact shape of the bug: a module-level import missing, the name read inside a function."""
    p = tmp_path / "t.py"
    p.write_text("import numpy as np\n\ndef f(x):\n    from a import b\n    return b(x, mean=ocr_host.DET_MEAN) + np.pi\n")
    top = symtable.symtable(p.read_text(), str(p), "exec")
    assert top is not None
    global ROOT
    keep, ROOT = ROOT, tmp_path
    try:
        assert undefined_names(p) == ["t.py:3 f(): name 'ocr_host' is not defined"]
    finally:
        ROOT = keep


def _repo_module_refs(path: Path):
    """(line, module, attribute) for every `from <repo module> import name` and every `alias.attr` where alias is bound by an
    import of a repo module (rapiddoc_amd.*, oracle.*), anywhere in the file (function-level imports included)."""
    import ast
    tree = ast.parse(path.read_text(), str(path))
    repo = ("rapiddoc_amd", "oracle")
    alias_of, refs = {}, []
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and node.module and node.level == 0 and node.module.split(".")[0] in repo:
            for a in node.names:
                refs.append((node.lineno, node.module, a.name))
                alias_of.setdefault(a.asname or a.name, set()).add(f"{node.module}.{a.name}")
        elif isinstance(node, ast.Import):
            for a in node.names:
                if a.name.split(".")[0] in repo and a.asname:
                    alias_of.setdefault(a.asname, set()).add(a.name)
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and len(alias_of.get(node.value.id, ())) == 1:
            refs.append((node.lineno, next(iter(alias_of[node.value.id])), node.attr))
    return refs


@pytest.mark.parametrize("path", FILES, ids=lambda p: str(p.relative_to(ROOT)))
def test_repo_module_attributes_exist(path):
    """`ocr_host.DET_MEAN`, `from rapiddoc_amd.pipeline import PagePipeline`, ...: the attribute exists on the imported module
    (modules only - an alias bound to a class or function is skipped)."""
    import importlib
    import types
    bad = []
    for line, modname, attr in _repo_module_refs(path):
        try:
            mod = importlib.import_module(modname)
        except ImportError:
            parent, _, leaf = modname.rpartition(".")
            obj = getattr(importlib.import_module(parent), leaf, None) if parent else None
            if obj is None:
                bad.append(f"{path.relative_to(ROOT)}:{line} cannot import {modname}")
            continue                                          # alias of a class / function / constant: not a module
        if not isinstance(mod, types.ModuleType):
            continue
        if not hasattr(mod, attr):
            try:
                importlib.import_module(f"{modname}.{attr}")  # `from package import submodule`
            except ImportError:
                bad.append(f"{path.relative_to(ROOT)}:{line} module {modname} has no attribute '{attr}'")
    assert bad == []
