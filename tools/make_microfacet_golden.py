"""Extract the golden VECTORS (numeric array literals: Mitsuba 0.6 reference data) of the reference's
src/librender/tests/test_microfacet.py into tests/golden/microfacet_vectors.json.  Only numbers are copied, keyed by the
test function they appear in, in source order; tests/test_oracle_known_answers.py rebuilds the inputs those tests
describe (theta / phi sweeps, 6 x 6 sample grids) and checks the oracle's MicrofacetDistribution against them.
Run in the build container (needs /root/reference); the GPU box only sees the committed JSON."""
import ast
import json
import os
import sys

SRC = "/root/reference/src/librender/tests/test_microfacet.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "microfacet_vectors.json")


def number(node):
    if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)):
        return float(node.value)
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, ast.USub):
        v = number(node.operand)
        return None if v is None else -v
    return None


def literal(node):
    """nested list of numbers, or None"""
    if isinstance(node, (ast.List, ast.Tuple)):
        out = []
        for e in node.elts:
            v = number(e)
            if v is None:
                v = literal(e)
            if v is None:
                return None
            out.append(v)
        return out if out else None
    return None


class Collect(ast.NodeVisitor):
    def __init__(self):
        self.found = []

    def visit_List(self, node):
        v = literal(node)
        if v is not None and len(v) >= 6:
            self.found.append(v)
        else:
            self.generic_visit(node)


tree = ast.parse(open(SRC).read())
out = {}
for fn in tree.body:
    if isinstance(fn, ast.FunctionDef) and fn.name.startswith("test0") and "chi2" not in fn.name and "construct" not in fn.name:
        c = Collect()
        c.visit(fn)
        out[fn.name] = c.found
json.dump({"source": "src/librender/tests/test_microfacet.py (numeric literals only)", "vectors": out}, open(OUT, "w"), indent=0)
for k, v in out.items():
    print(k, [(len(a), len(a[0]) if isinstance(a[0], list) else 1) for a in v], file=sys.stderr)
