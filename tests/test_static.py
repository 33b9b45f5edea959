"""Static guard for the code paths the CPU suite cannot execute (hipGraph capture, multi-GPU bench legs): every global
name the product modules, bench.py and __graft_entry__.py load must be defined somewhere in that module."""
import ast
import builtins
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _undefined(path):
    tree = ast.parse(open(path).read())
    defined = set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
    for n in ast.walk(tree):
        if isinstance(n, (ast.Import, ast.ImportFrom)):
            for a in n.names:
                defined.add((a.asname or a.name).split(".")[0])
        elif isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            defined.add(n.name)
            if not isinstance(n, ast.ClassDef):
                a = n.args
                for x in a.posonlyargs + a.args + a.kwonlyargs:
                    defined.add(x.arg)
                for x in (a.vararg, a.kwarg):
                    if x is not None:
                        defined.add(x.arg)
        elif isinstance(n, ast.Lambda):
            for x in n.args.args:
                defined.add(x.arg)
        elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            defined.add(n.id)
        elif isinstance(n, ast.ExceptHandler) and n.name:
            defined.add(n.name)
    return sorted({n.id for n in ast.walk(tree) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in defined})


def test_no_undefined_names():
    files = glob.glob(os.path.join(ROOT, "cc_amd", "*.py")) + glob.glob(os.path.join(ROOT, "cc_amd", "models", "*.py")) + \
        [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    bad = {os.path.relpath(f, ROOT): u for f in files for u in [_undefined(f)] if u}
    assert not bad, bad
