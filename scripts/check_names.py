#!/usr/bin/env python3
"""Names a module loads but never binds (a missing import after code moved between files): conservative -- any binding anywhere in
the module counts.  python3 scripts/check_names.py benchlib/*.py"""
import ast
import builtins
import sys

bad = 0
for path in sys.argv[1:]:
    tree = ast.parse(open(path).read(), path)
    bound = set(dir(builtins)) | {'__file__', '__name__'}
    for node in ast.walk(tree):
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            for a in node.names:
                bound.add((a.asname or a.name).split(".")[0])
        elif isinstance(node, (ast.FunctionDef, ast.ClassDef, ast.AsyncFunctionDef)):
            bound.add(node.name)
            if not isinstance(node, ast.ClassDef):
                for a in node.args.args + node.args.kwonlyargs + getattr(node.args, "posonlyargs", []):
                    bound.add(a.arg)
                for a in (node.args.vararg, node.args.kwarg):
                    if a:
                        bound.add(a.arg)
        elif isinstance(node, ast.Lambda):
            for a in node.args.args + node.args.kwonlyargs:
                bound.add(a.arg)
        elif isinstance(node, ast.Name) and isinstance(node.ctx, (ast.Store, ast.Del)):
            bound.add(node.id)
        elif isinstance(node, ast.ExceptHandler) and node.name:
            bound.add(node.name)
    for node in ast.walk(tree):
        if isinstance(node, ast.Name) and isinstance(node.ctx, ast.Load) and node.id not in bound:
            print("%s:%d: undefined name %r" % (path, node.lineno, node.id))
            bad += 1
sys.exit(1 if bad else 0)
