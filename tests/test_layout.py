"""Repository rules: the product never touches the oracle or a CPU fallback; the C-ABI boundary and
required top-level files exist."""
import ast
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "karanta_ocr_amd")


def py_files(d):
    for base, _, files in os.walk(d):
        for f in files:
            if f.endswith(".py"):
                yield os.path.join(base, f)


def test_product_never_imports_the_oracle():
    for path in py_files(PKG):
        tree = ast.parse(open(path).read())
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            for n in names:
                assert not n.split(".")[0] == "oracle", f"{path} imports {n}"
        assert "oracle" not in re.findall(r"importlib\.import_module\(['\"](\w+)", open(path).read())


def test_only_checker_sites_import_the_oracle():
    allowed = {"bench.py", "__graft_entry__.py"}
    for f in os.listdir(ROOT):
        if f.endswith(".py") and f not in allowed:
            assert "from oracle" not in open(os.path.join(ROOT, f)).read(), f
    src = open(os.path.join(ROOT, "bench.py")).read()
    # in bench.py the oracle's ARITHMETIC appears only inside cpu_baseline() (the parity legs consume its run); the one other
    # import is the module-level table of stated tolerances (oracle/tolerances.py: constants, no computation)
    tree = ast.parse(src)
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef):
            uses = any(isinstance(n, ast.ImportFrom) and (n.module or "").startswith("oracle") for n in ast.walk(node))
            assert uses == (node.name == "cpu_baseline"), node.name
    top = [n.module for n in tree.body if isinstance(n, ast.ImportFrom) and (n.module or "").startswith("oracle")]
    assert top == ["oracle.tolerances"], top


def test_engine_has_no_cpu_fallback():
    src = open(os.path.join(PKG, "engine.py")).read()
    assert "no CPU fallback" in src and "torch.matmul" not in src and "F.scaled_dot_product_attention" not in src
    for path in py_files(PKG):
        s = open(path).read()
        assert "torch.nn.functional" not in s and "import torch.nn" not in s, path


def test_required_files_exist():
    for f in ("include/karanta_hip.h", "bench.py", "__graft_entry__.py", "DESIGN.md", "INTEGRATION.md", "oracle/qwen2vl_oracle.py",
              "tests/golden/make_golden.py", "tests/golden/qwen2vl_tiny_golden.npz"):
        assert os.path.exists(os.path.join(ROOT, f)), f
    assert os.path.isdir(os.path.join(ROOT, "profiles"))


def test_no_reference_sources_in_repo():
    """Fixtures are data; no file of the reference is stored under tests/ in any encoding."""
    for base, _, files in os.walk(os.path.join(ROOT, "tests")):
        for f in files:
            assert not f.endswith((".sh", ".yaml", ".toml")), f
