"""Extracts (module, function, [parameter names]) from the reference's type stubs
hydrainfer/_C/**/__init__.pyi into tests/golden/op_signatures.json (data only: names and arity).
Run in the build container:  python tests/golden/generate_stub_signatures.py"""
import ast
import json
import os

REFERENCE = os.environ.get("HYDRA_REFERENCE", "/root/reference")
root = os.path.join(REFERENCE, "hydrainfer", "_C")
out = {}
for dirpath, _, files in os.walk(root):
    for f in files:
        if f != "__init__.pyi":
            continue
        rel = os.path.relpath(dirpath, os.path.join(REFERENCE, "hydrainfer")).replace(os.sep, ".")
        tree = ast.parse(open(os.path.join(dirpath, f)).read())
        fns = {n.name: [a.arg for a in n.args.args] for n in tree.body if isinstance(n, ast.FunctionDef)}
        if fns:
            out[rel] = fns
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "op_signatures.json")
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
print({k: sorted(v) for k, v in out.items()})
