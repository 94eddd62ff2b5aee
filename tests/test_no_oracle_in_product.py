"""The oracle is test infrastructure: nothing in the product package may import, load or execute it."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRODUCT = os.path.join(ROOT, "nsdp_amd")
# importing the package `oracle`, or reaching its files by path
PATTERNS = [re.compile(r"^\s*(from|import)\s+oracle\b", re.M), re.compile(r"\boracle[/\\.](_build|_ref|tdnet_ref|pointnet2_ref)"),
            re.compile(r"libnsdp_oracle"), re.compile(r"import_module\(\s*['\"]oracle")]


def _sources():
    for base, _, files in os.walk(PRODUCT):
        if os.path.basename(base) in ("lib", "obj", "__pycache__"):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")):
                yield os.path.join(base, f)


def test_product_package_never_touches_the_oracle():
    offenders = []
    n = 0
    for path in _sources():
        n += 1
        with open(path, encoding="utf-8", errors="replace") as fh:
            text = fh.read()
        for pat in PATTERNS:
            m = pat.search(text)
            if m:
                offenders.append((os.path.relpath(path, ROOT), m.group(0).strip()))
    assert n > 30
    assert not offenders, offenders


def test_only_the_allowed_files_import_the_oracle():
    """Outside tests/ and oracle/ itself: bench.py (cpu_baseline leg) and smoke_model.py (called by
    __graft_entry__.smoke()); __graft_entry__.py builds and smoke-checks with it."""
    allowed = {"bench.py", "smoke_model.py", "__graft_entry__.py"}
    found = set()
    for f in os.listdir(ROOT):
        if f.endswith(".py"):
            with open(os.path.join(ROOT, f)) as fh:
                if PATTERNS[0].search(fh.read()):
                    found.add(f)
    for base, _, files in os.walk(os.path.join(ROOT, "tools")):
        for f in files:
            if f.endswith(".py"):
                with open(os.path.join(base, f)) as fh:
                    assert not PATTERNS[0].search(fh.read()), f
    assert found <= allowed, found - allowed
